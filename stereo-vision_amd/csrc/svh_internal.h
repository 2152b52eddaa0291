// Internal declarations shared by the host stages, the HIP kernels and the
// engine of libsvhip.so.  Nothing here is part of the public C-ABI.
#ifndef SVH_INTERNAL_H
#define SVH_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/svh.h"

namespace svh {

// ---------------------------------------------------------------- geometry
struct Dims {
    int32_t W, H;        // image
    int32_t DW, DH;      // disparity map (W/2,H/2 when subsampling)
    int32_t step;        // candidate lattice step (elas.cpp:453-457)
    int32_t Wc, Hc;      // candidate lattice (elas.cpp:460-463)
    int32_t gw, gh;      // disparity grid (elas.cpp:98-99)
};
Dims make_dims(const svh_elas_params& p, int32_t W, int32_t H);

// One rasterisation record per triangle and image side, prepared on the host
// with the reference's float arithmetic (elas.cpp:1026-1072) so that span
// boundaries cannot differ; the device only evaluates a*u+b per column.
struct TriRaster {
    int32_t uA, uB, uC;        // (int32)A_u, (int32)B_u, (int32)C_u
    float ACa, ACb;            // long edge
    float ABa, ABb;            // first part
    float BCa, BCb;            // second part
    float pa, pb, pc;          // disparity plane of this side
    int32_t valid;             // |plane_a|<0.7 && |plane_d|<0.7
};

// Host-side result of stages E4(second half)..E9 for one pair.
struct HostPrior {
    std::vector<int32_t> support;            // n x (u,v,d)
    std::vector<int32_t> tri[2];             // n x 3
    std::vector<float> planes[2];            // n x 6 (t1a,t1b,t1c,t2a,t2b,t2c)
    std::vector<TriRaster> raster[2];
    std::vector<int32_t> cell_off[2];        // cells+1 prefix offsets into cell_d
    std::vector<uint16_t> cell_d[2];         // ascending disparities per cell
    std::vector<int32_t> P;                  // prior table (elas.cpp:984-992)
    int32_t plane_radius;
};

// E5/E6 + list + corners (elas.cpp:174-318, 495-523); dcan is modified in place
void support_from_candidates(const svh_elas_params& p, const Dims& d, int16_t* dcan,
                             std::vector<int32_t>& support);
// E7 (both sides), E8, raster records, E9, prior table.  Returns false when a
// triangulation fails.
bool build_prior(const svh_elas_params& p, const Dims& d, HostPrior& hp);
// reference-layout grid (int32 [gh][gw][disp_max+2]) for the stage tap
void expand_grid(const svh_elas_params& p, const Dims& d, const HostPrior& hp, int side,
                 std::vector<int32_t>& grid);

int32_t delaunay(const float* pts, int32_t n, int32_t* tri, int32_t cap);

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
};
struct DevImages {
    const uint8_t* I[2];   // left, right
    int32_t pitch[2];
};

void launch_descriptor(const LaunchCtx& cx, const DevImages& img, int32_t W, int32_t H, int32_t half,
                       uint8_t* desc1, uint8_t* desc2);
void launch_support(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, const uint8_t* desc1,
                    const uint8_t* desc2, int16_t* dcan);
void launch_owner(const LaunchCtx& cx, const Dims& d, const TriRaster* r1, int32_t n1, const TriRaster* r2,
                  int32_t n2, int32_t subsampling, int32_t* owner1, int32_t* owner2);
struct MatchArgs {
    const uint8_t* desc[2];
    const int32_t* owner[2];
    const TriRaster* raster[2];
    const int32_t* cell_off[2];
    const uint16_t* cell_d[2];
    const int32_t* P;          // device prior table
    float* D[2];
    int32_t plane_radius;
};
void launch_match(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, const MatchArgs& a);
void launch_lr(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, const float* D1raw,
               const float* D2raw, float* D1, float* D2);
// speckle removal: labels/runlen/counts are scratch of DW*DH int32 each
void launch_segments(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, float* D,
                     int32_t* labels, int32_t* runlen, int32_t* counts);
void launch_gap(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, float* D, float* tmp);
void launch_adaptive_mean(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, float* D,
                          float* tmp);
void launch_median(const LaunchCtx& cx, const Dims& d, float* D, float* tmp);

}  // namespace svh
#endif
