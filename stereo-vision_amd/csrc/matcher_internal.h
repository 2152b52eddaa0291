// Internal declarations of the Matcher path (kernels <-> engine).
#ifndef SVH_MATCHER_INTERNAL_H
#define SVH_MATCHER_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <functional>

#include "../../include/svh.h"
#include "svh_config.h"

namespace svh {

// a feature table on the device with its bin index (CSR over class x v_bin x u_bin)
struct FeatView {
    const int32_t* rec;     // [n][12]: u, v, 0, class, d1..d8
    const int32_t* count;   // device-resident n
    const int32_t* off;     // [4*vb*ub + 1]
    const int32_t* ids;     // feature indices, ascending inside each bin
};

// Sobel images used by the refinement (full resolution when half_resolution is on)
struct SobelView {
    const uint8_t* du;
    const uint8_t* dv;
    int32_t w, h, bpl;
};

struct MatchParams {
    int32_t method;
    int32_t width, height;        // current frame (bin grid, pixel-owner map)
    int32_t ub, vb, binsize;
    int32_t match_radius, match_disp_tolerance;
    int32_t has_tr;
    double f, cu, cv, base;
    double tr[12];                // rows 0..2 of Tr_delta
};

void mlaunch_upload(void* stream, const uint8_t* pinned, uint8_t* dev, size_t bytes);
// small transfers (a multiple of 4 bytes) between pinned host and device memory: hipMemcpyAsync / hipMemsetAsync,
// or -- while a batch is being recorded (batch_rec.h) -- jobs of one copy / fill kernel per phase
void mlaunch_copy(void* stream, void* dst, const void* src, size_t bytes, int kind);
void mlaunch_fill(void* stream, void* dst, int byte_value, size_t bytes);
void mlaunch_half(void* stream, const uint8_t* I, int bpl, uint8_t* out, int hw, int hh, int hbpl);
// f1 == nullptr: Sobel only
void mlaunch_filters(void* stream, const uint8_t* I, int w, int h, int bpl, uint8_t* du, uint8_t* dv,
                     int16_t* f1, int16_t* f2);
int  mnms_blocks(int extent, int n, int margin);
void mlaunch_half_filters(void* stream, const uint8_t* I, int w, int h, int bpl, uint8_t* Ih, int hw, int hh, int hbpl,
                          uint8_t* du_full, uint8_t* dv_full);
void mlaunch_features(void* stream, const int16_t* f1, const int16_t* f2, const uint8_t* du,
                      const uint8_t* dv, int w, int h, int bpl, int n, int tau, int margin, int scale,
                      int4* slots, int32_t* flags, int32_t* order, int32_t* table, int32_t* count);
void mlaunch_features2(void* stream, const int16_t* f1, const int16_t* f2, const uint8_t* du, const uint8_t* dv, int w,
                       int h, int bpl, int tau, int margin, int scale, int n_a, int4* slots_a, int32_t* flags_a,
                       int32_t* order_a, int32_t* table_a, int32_t* count_a, int n_b, int4* slots_b, int32_t* flags_b,
                       int32_t* order_b, int32_t* table_b, int32_t* count_b, int32_t* host_counts);
// up to 8 feature tables whose bin indices are built by one launch (one workgroup each)
struct BinJobs {
    const int32_t* table[8];
    const int32_t* count[8];
    int32_t* off[8];
    int32_t* ids[8];
};
// n_host_max: the largest feature count among the jobs as known on the host (sizes the LDS build)
void mlaunch_bin_index(void* stream, const BinJobs& J, int njobs, int n_host_max, int ub, int vb, int binsize,
                       int32_t* cursor);
void mlaunch_match(void* stream, const MatchParams& P, const FeatView& m1p, const FeatView& m2p,
                   const FeatView& m1c, const FeatView& m2c, int nquery_cap, const float* ranges,
                   int use_prior, svh_p_match* slots, int32_t* flags, int32_t* pixel_owner,
                   svh_p_match* out, int32_t* out_count, int32_t* out_count_host = nullptr);
// parabolic = 0: relocateMinimum in place; 1: parabolicFitting, survivors compacted into
// `compacted` / `compacted_count`
void mlaunch_refine(void* stream, svh_p_match* m, const int32_t* count, int cap, int method, int margin,
                    const SobelView& s1p, const SobelView& s2p, const SobelView& s1c,
                    const SobelView& s2c, int parabolic, int32_t* flags, svh_p_match* compacted,
                    int32_t* compacted_count);

// par_depth > 0: the top `par_depth` levels of the divide-and-conquer run their halves on two
// threads (2^par_depth threads in all); the output is identical to the sequential run
int32_t delaunay(const float* pts, int32_t n, int32_t* tri, int32_t cap, int par_depth = 0, bool expect_dups = false);
// the helper threads of delaunay.cpp (see svh_internal.h): fn(0) here, fn(1 .. k-1) on helpers; poll / sleep
void run_many(int k, const std::function<void(int)>& fn);
void helpers_warm(int want, int us);

}  // namespace svh
#endif
