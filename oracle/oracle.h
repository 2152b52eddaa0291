/*
 * TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the reference hot path.
 *
 * Plain scalar C++ (no SSE, no HIP), built with -ffp-contract=off so float
 * expressions round exactly like the reference's -msse3 build.  Every function
 * cites the reference lines it follows (paths relative to the reference
 * checkout).  Only tests/, bench.py's cpu_baseline leg and
 * __graft_entry__.smoke() may load liboracle.so; the product never does.
 *
 * Pinning: checked bit-for-bit against the real reference (oracle/_ref, built
 * from /root/reference by oracle/Makefile) and against the golden fixtures in
 * tests/golden/ that were generated from it (tests/test_oracle_*.py).
 *
 * The one stage NOT restated here is Shewchuk's Triangle (libelas/src/
 * triangle.cpp): the oracle takes the triangulation as an input (callback or
 * fixture), supplied by the real Triangle in oracle/_ref or by golden triangle
 * lists.  The product's own host triangulator (svh_delaunay) is compared
 * against those directly.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#include "../include/svh.h"
#include "../include/svh_map.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- libelas stages ------------------------------------------------------ */
/* filter::sobel3x3 (libelas/src/filter.cpp:408-416): defined on rows 1..h-2,
 * cols 1..w-2 of the visible image (others written 0). */
void orc_sobel3x3(const uint8_t* I, uint8_t* du, uint8_t* dv, int32_t w, int32_t h, int32_t bpl);
/* Descriptor::createDescriptor (descriptor.cpp:48-121); border = 0. */
void orc_descriptor(const uint8_t* I, int32_t w, int32_t h, int32_t bpl, int32_t half,
                    uint8_t* desc);
/* candidate lattice size (elas.cpp:453-463) */
void orc_dcan_dims(const svh_elas_params* p, int32_t w, int32_t h, int32_t* wc, int32_t* hc);
/* forward/backward candidate matching (elas.cpp:322-445, 464-493) */
void orc_support_candidates(const svh_elas_params* p, const uint8_t* desc1, const uint8_t* desc2,
                            int32_t w, int32_t h, int16_t* dcan);
/* in-place filters + list (elas.cpp:174-279, 495-523, 283-318); returns n */
int32_t orc_support_filter(const svh_elas_params* p, int16_t* dcan, int32_t w, int32_t h,
                           int32_t* support, int32_t cap);
/* computeDisparityPlanes (elas.cpp:605-680) with Matrix::solve (matrix.cpp:414-501) */
void orc_planes(const int32_t* support, const int32_t* tri, int32_t ntri, float* planes);
/* createGrid (elas.cpp:684-780) -> int32 [gh][gw][disp_max+2] */
void orc_grid_dims(const svh_elas_params* p, int32_t w, int32_t h, int32_t* gw, int32_t* gh);
void orc_grid(const svh_elas_params* p, const int32_t* support, int32_t nsup, int32_t w, int32_t h,
              int32_t right, int32_t* grid);
/* computeDisparity + findMatch (elas.cpp:814-1118) */
void orc_dense(const svh_elas_params* p, const int32_t* support, const int32_t* tri,
               const float* planes, int32_t ntri, const int32_t* grid, const uint8_t* desc1,
               const uint8_t* desc2, int32_t w, int32_t h, int32_t right, float* D);
/* post-processing (elas.cpp:1122-1838); dw/dh = disparity map dims */
void orc_lr_check(const svh_elas_params* p, float* D1, float* D2, int32_t dw, int32_t dh);
void orc_remove_small_segments(const svh_elas_params* p, float* D, int32_t dw, int32_t dh);
void orc_gap_interpolation(const svh_elas_params* p, float* D, int32_t dw, int32_t dh);
void orc_adaptive_mean(const svh_elas_params* p, float* D, int32_t dw, int32_t dh);
void orc_median(const svh_elas_params* p, float* D, int32_t dw, int32_t dh);

/* triangulator supplied by the caller: n points (x,y) -> up to cap triangles,
 * returns the count (see header comment). */
typedef int32_t (*orc_triangulate_fn)(const float* pts, int32_t n, int32_t* tri, int32_t cap);

/* Elas::process (elas.cpp:32-170) end to end, keeping every intermediate. */
typedef struct orc_run orc_run;
orc_run* orc_elas_run(const svh_elas_params* p, const uint8_t* I1, const uint8_t* I2,
                      const int32_t* dims, orc_triangulate_fn tri_fn);
int32_t  orc_elas_run_status(orc_run* r);
int64_t  orc_elas_run_get(orc_run* r, int32_t stage, void* buf, int64_t cap);
void     orc_elas_run_free(orc_run* r);
/* final maps only; returns status (0 ok, 1 = <3 support points, D untouched) */
int32_t  orc_elas_process(const svh_elas_params* p, const uint8_t* I1, const uint8_t* I2,
                          float* D1, float* D2, const int32_t* dims, orc_triangulate_fn tri_fn);

/* ---- libviso2 Matcher (oracle/viso_oracle.cpp) ---------------------------- */
/* Same call surface as the svh_matcher_* C-ABI (include/svh.h); the Delaunay
 * triangulator of removeOutliers is supplied by the caller. */
typedef struct orc_matcher orc_matcher;
void         orc_matcher_params_default(svh_matcher_params* p);
orc_matcher* orc_matcher_create(const svh_matcher_params* p);
void         orc_matcher_destroy(orc_matcher* m);
void         orc_matcher_set_triangulator(orc_matcher* m, orc_triangulate_fn fn);
void         orc_matcher_set_intrinsics(orc_matcher* m, double f, double cu, double cv, double base);
int32_t      orc_matcher_push_back(orc_matcher* m, const uint8_t* I1, const uint8_t* I2,
                                   const int32_t* dims, int32_t replace);
int32_t      orc_matcher_match_features(orc_matcher* m, int32_t method, const double* Tr_delta);
int32_t      orc_matcher_bucket_features(orc_matcher* m, int32_t max_features, float bw, float bh);
float        orc_matcher_get_gain(orc_matcher* m, const int32_t* inliers, int32_t n);
int32_t      orc_matcher_get_matches(orc_matcher* m, svh_p_match* out, int32_t cap);
int32_t      orc_matcher_get_features(orc_matcher* m, int32_t table, int32_t* out, int32_t cap);
int64_t      orc_matcher_get_stage(orc_matcher* m, int32_t stage, void* buf, int64_t cap);
int64_t      orc_matcher_get_filter(orc_matcher* m, int32_t which, void* buf, int64_t cap,
                                    int32_t* dims3);

/* ---- libviso2 VisualOdometryStereo (oracle/viso_oracle.cpp) ------------------ */
/* Same call surface as the svh_vo_* C-ABI (include/svh.h). */
typedef struct orc_vo orc_vo;
void    orc_vo_params_default(svh_vo_params* p);
orc_vo* orc_vo_create(const svh_vo_params* p);
void    orc_vo_destroy(orc_vo* v);
void    orc_vo_set_triangulator(orc_vo* v, orc_triangulate_fn fn);
int32_t orc_vo_process(orc_vo* v, const uint8_t* I1, const uint8_t* I2, const int32_t* dims, int32_t replace);
int32_t orc_vo_process_matches(orc_vo* v, const svh_p_match* m, int32_t n);
int32_t orc_vo_estimate_motion(orc_vo* v, const svh_p_match* m, int32_t n, double* tr6);
void    orc_vo_get_motion(orc_vo* v, double* Tr16);
int32_t orc_vo_get_inliers(orc_vo* v, int32_t* out, int32_t cap);
int32_t orc_vo_num_matches(orc_vo* v);
int32_t orc_vo_get_matches(orc_vo* v, svh_p_match* out, int32_t cap);
float   orc_vo_get_gain(orc_vo* v, const int32_t* inliers, int32_t n);

/* ---- map fusion (stereomapper/stereothread.cpp:180-437), see map_oracle.cpp ---------------- */
typedef svh_map_params orc_map_params;
typedef struct orc_map orc_map;
void     orc_map_coeffs(const orc_map_params* p, const double* H16, float* hcf12, float* hfc4, float* pfc12);
orc_map* orc_map_create(const orc_map_params* p);
void     orc_map_destroy(orc_map* m);
void     orc_map_clear(orc_map* m);
void     orc_map_add(orc_map* m, const float* D1, const uint8_t* I1, const int32_t* dims, const double* H16,
                     float gain);
int64_t  orc_map_points(const orc_map* m, int32_t which, float* xyzv, int64_t cap);
void     orc_map_planes(const orc_map* m, float* out5);
void     orc_disparity_colormap(const float* D, int64_t n, float* rgb);

#ifdef __cplusplus
}
#endif
#endif
