/* The three C-ABI headers are plain C: this file is compiled as C99 with -pedantic and linked
 * against libsvhip.so.  Usage: capi_c99 <png> <expected_width> <expected_height>
 * Exercises only host-side entries (it must also run on a machine without a GPU). */
#include <stdio.h>
#include <stdlib.h>

#include "svh.h"
#include "svh_kitti.h"
#include "svh_map.h"

int main(int argc, char** argv) {
    svh_elas_params ep;
    svh_matcher_params mp;
    svh_vo_params vp;
    svh_map_params fp = {645.24f, 635.96f, 194.13f, 0.5707f, 20.0f};
    int32_t w = 0, h = 0;
    svh_elas_params_default(&ep, SVH_ELAS_ROBOTICS);
    svh_matcher_params_default(&mp);
    svh_vo_params_default(&vp);
    if (ep.disp_max != 255 || mp.nms_n != 3 || vp.ransac_iters != 200) return 2;
    if (argc >= 4) {
        uint8_t* img;
        if (svh_png_read_gray(argv[1], NULL, 0, &w, &h) != SVH_OK) return 3;
        if (w != atoi(argv[2]) || h != atoi(argv[3])) return 4;
        img = (uint8_t*)malloc((size_t)w * h);
        if (svh_png_read_gray(argv[1], img, (size_t)w * h, &w, &h) != SVH_OK) return 5;
        printf("%d %d %d\n", w, h, (int)img[0]);
        free(img);
    }
    if (svh_device_count() < 1) {
        /* no CPU fallback: the device objects refuse to exist */
        if (svh_map_create(&fp) != NULL) return 6;
    } else {
        svh_map* m = svh_map_create(&fp);
        if (!m) return 7;
        svh_map_destroy(m);
    }
    printf("capi ok: %s\n", svh_version());
    return 0;
}
