/*
 * svh_kitti.h -- the on-disk formats in front of the hot path: a KITTI raw
 * drive as stereomapper's file-playback thread reads it, without OpenCV or Qt.
 *
 *   <drive>/image_00/data/%010d.png, image_00/timestamps.txt     left  gray camera
 *   <drive>/image_01/data/%010d.png, image_01/timestamps.txt     right gray camera
 *   <calib dir>/calib_cam_to_cam.txt                             rectified projections
 *
 * Host-only entry points of libsvhip.so (no device work); plain C like svh.h.
 * Reference interfaces (paths relative to the reference checkout):
 *   stereomapper/calibiokitti.cpp:227-262   readCamToCamCalibFromFile
 *   stereomapper/calibiokitti.cpp:176-224   readCalibFileMatrix (values parsed as float)
 *   stereomapper/stereothread.cpp:444-447   f, cu, cv, base from P_rect_00 / P_rect_01
 *   stereomapper/stereoimageiokitti.cpp:81-119  getNextImageDataSet (frame numbering from 0,
 *                                           time of day from the timestamp line, gray load)
 */
#ifndef SVH_KITTI_H
#define SVH_KITTI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVH_KITTI_CAMERAS 4

/* calib_cam_to_cam.txt, every matrix row-major (calibiokitti.h: _cam_to_cam_*) */
typedef struct svh_kitti_calib {
    char   calib_time[64];
    double corner_dist;
    double S[SVH_KITTI_CAMERAS][2];
    double K[SVH_KITTI_CAMERAS][9];
    double D[SVH_KITTI_CAMERAS][5];
    double R[SVH_KITTI_CAMERAS][9];
    double T[SVH_KITTI_CAMERAS][3];
    double S_rect[SVH_KITTI_CAMERAS][2];
    double R_rect[SVH_KITTI_CAMERAS][9];
    double P_rect[SVH_KITTI_CAMERAS][12];
    /* the gray stereo rig as StereoThread derives it (stereothread.cpp:444-447):
     * f = P_rect_00[0][0], cu = P_rect_00[0][2], cv = P_rect_00[1][2],
     * base = -P_rect_01[0][3] / P_rect_01[0][0]                                         */
    double f, cu, cv, base;
} svh_kitti_calib;

/* 0 on success; SVH_ERR_BAD_ARG when the file is missing or an entry is absent or has the
 * wrong element count (the reference prints an error and returns false there).           */
int32_t svh_kitti_read_cam_to_cam(const char* path, svh_kitti_calib* out);

/* One PNG as an 8-bit gray image, rows packed (the reference loads with
 * CV_LOAD_IMAGE_GRAYSCALE, stereoimageiokitti.cpp:111).  Non-interlaced PNG of colour type
 * gray, gray+alpha, RGB or RGBA, 8 or 16 bits per sample; colour is reduced with OpenCV's
 * fixed-point weights (R*4899 + G*9617 + B*1868 + 8192) >> 14, 16-bit samples keep their
 * high byte.  buf may be NULL to query the size.  Returns 0, SVH_ERR_BAD_ARG (unreadable,
 * not a PNG, cap too small) or SVH_ERR_UNSUPPORTED (interlaced, palette, other depths).   */
int32_t svh_png_read_gray(const char* path, uint8_t* buf, size_t cap, int32_t* width, int32_t* height);

/* A drive directory played back frame by frame (StereoImageIOKITTI::setUpDataPath +
 * getNextImageDataSet).                                                                   */
typedef struct svh_kitti_seq svh_kitti_seq;
svh_kitti_seq* svh_kitti_seq_open(const char* drive_dir);   /* NULL: directory or timestamps missing */
void           svh_kitti_seq_close(svh_kitti_seq* s);
int32_t        svh_kitti_seq_count(const svh_kitti_seq* s); /* frames = timestamp lines of both cameras */
/* The frame the next call returns (0 .. count): a rank of a sharded run starts at its share. */
int32_t        svh_kitti_seq_seek(svh_kitti_seq* s, int32_t frame);
/* Next stereo pair.  I1/I2 receive width*height bytes each (cap = bytes available in each);
 * dims = {width, height, width}.  tv = {left sec, left usec, right sec, right usec}: time of
 * day, hour*3600 + minute*60 + second and nanoseconds/1000 (stereoimageiokitti.cpp:100-105).
 * Returns 0, 1 at the end of the sequence, or a negative SVH_ERR_* (the frame is consumed).  */
int32_t svh_kitti_seq_next(svh_kitti_seq* s, uint8_t* I1, uint8_t* I2, size_t cap, int32_t* dims,
                           int64_t* tv);

#ifdef __cplusplus
}
#endif
#endif
