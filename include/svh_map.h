/*
 * svh_map.h -- the consumer of D1 after the hot path: stereomapper's 3-D
 * reprojection and frame-to-frame map fusion on the device.
 *
 * Reference interfaces (paths relative to the reference checkout):
 *   stereomapper/stereothread.cpp:180-255   StereoThread::createCurrentMap
 *   stereomapper/stereothread.cpp:290-437   StereoThread::addDisparityMapToReconstruction
 *   stereomapper/stereothread.cpp:441-456   getIntrinsics (f, cu, cv, base, K)
 *   stereomapper/stereothread.cpp:460-470   clearReconstruction
 *
 * The reference keeps `_previous_map3d` pointing at buffers it has just freed (:432-433); this
 * library implements the intended behaviour -- the previous map is the current map of the frame
 * before, as the fusion left it -- see DESIGN.md.  Plain C like svh.h; device work runs on the
 * object's own stream and every call returns when it is complete.
 */
#ifndef SVH_MAP_H
#define SVH_MAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the float members StereoThread reads its geometry from (stereothread.h:192-196) */
typedef struct svh_map_params {
    float f, cu, cv, base;
    float max_dist;              /* StereoThread::_max_dist, 20 in the constructor (:14) */
} svh_map_params;

typedef struct svh_map svh_map;

svh_map* svh_map_create(const svh_map_params* p);   /* NULL without a HIP device */
void     svh_map_destroy(svh_map* m);
/* clearReconstruction(): forget the previous map and the point lists */
void     svh_map_clear(svh_map* m);

/* One frame = pushBack(simage, H_total, gain) + the "reconstruction" step of run()
 * (stereothread.cpp:30-41, 166-170).
 *   D1        left disparity map, dims[0] x dims[1] floats, rows packed: a host pointer, or -- with
 *             d1_on_device -- the device pointer Elas::process wrote to (no copy of the map then)
 *   I1        left image on the host, dims[2] bytes per row
 *   H_total   4x4 camera pose, row major (StereoThread::_H_total)
 *   gain      VisualOdometryStereo::getGain of the frame (0: no gain correction)
 * Returns SVH_OK or a negative SVH_ERR_*.                                                      */
int32_t svh_map_add(svh_map* m, const float* D1, int32_t d1_on_device, const uint8_t* I1,
                    const int32_t* dims, const double* H_total, float gain);

/* The two point lists StereoThread::_points holds after a frame, as (x, y, z, val) floats in the
 * reference's push_back order (columns left to right, each top to bottom):
 *   which 0   points of the previous map that were not merged into the current one
 *   which 1   points of the current map (after the fusion)
 * Copies up to cap points to `xyzv` (may be NULL) and returns the number of points.            */
int64_t svh_map_points(svh_map* m, int32_t which, float* xyzv, int64_t cap);

/* The colour-coded disparity map StereoThread shows next to the image (stereothread.cpp:117-147):
 * hue from red (near, d >= 200) over green to magenta (far), black where D <= 0.  D: n floats on the
 * host or (d_on_device) on the device; rgb receives 3*n floats on the host, interleaved.           */
int32_t svh_disparity_colormap(const float* D, int32_t d_on_device, int64_t n, float* rgb);

/* Test access: the current map's planes I, D, X, Y, Z (5 x width*height floats) after the frame. */
int32_t svh_map_planes(svh_map* m, float* out5, size_t cap_floats);

#ifdef __cplusplus
}
#endif
#endif
