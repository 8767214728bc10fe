// Trajectory-map rasteriser: the control maps the reference draws with cv2.line(thickness 3) + cv2.circle(radius 3,
// filled) per track and step (scripts/run_inference_vipseg_json_repro.py:435-444, utils/dataset.py:755-764), written directly as
// the [-1, 1] fp16 / fp32 tensor the ControlNet's condition encoder consumes.  One thread per pixel walks the tracks of its map
// in draw order (later primitives overwrite earlier ones, exactly like drawing into one image).  Integer arithmetic only.
// Primitives (OpenCV's, as remembered - the package is not in the image): line(thickness 3) = ThickLine: the rectangle of
// half-width thickness / 2 + 1 / 2 = 2.0 px around the segment + discs of radius 2 at both ends; filled circle of radius r = the
// midpoint rule, which for r <= 3 is dx^2 + dy^2 <= r^2.
#include "pt_common.h"

namespace {

// pts int32 [n_tracks][n_points][2]; map m draws, for every track in order, the segment point (start + m) -> (start + m + 1) in
// BGR (0, 0, 255) and the disc at point (start + m + 1) in (0, 255, 0); flip_mode 0: channels reversed once per map (inference
// script), 1: after EVERY track (the training dataset's cvtColor inside the loop: tracks alternate colours).
template <typename T>
__global__ __launch_bounds__(256) void rasterize_tracks_kernel(const int* __restrict__ pts, int n_tracks, int n_points, int start,
                                                               int n_maps, int n_total, int H, int W, int flip_mode,
                                                               T* __restrict__ out) {
    const int m = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (int64_t)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
        int c0 = 0, c1 = 0, c2 = 0;
        if (m < n_maps) {
            for (int t = 0; t < n_tracks; ++t) {
                const int* a = pts + ((int64_t)t * n_points + start + m) * 2;
                const long long x0 = a[0], y0 = a[1], x1 = a[2], y1 = a[3];
                const long long dx = x1 - x0, dy = y1 - y0, l2 = dx * dx + dy * dy;
                const long long vx = x - x0, vy = y - y0, ux = x - x1, uy = y - y1;
                bool on = vx * vx + vy * vy <= 4 || ux * ux + uy * uy <= 4;
                if (!on && l2 > 0) {
                    const long long cross = dx * vy - dy * vx, dot = dx * vx + dy * vy;
                    on = cross * cross <= 4 * l2 && dot >= 0 && dot <= l2;
                }
                if (on) { c0 = 0; c1 = 0; c2 = 255; }
                if (ux * ux + uy * uy <= 9) { c0 = 0; c1 = 255; c2 = 0; }
                if (flip_mode == 1) { const int s = c0; c0 = c2; c2 = s; }
            }
            if (flip_mode == 0) { const int s = c0; c0 = c2; c2 = s; }
        }
        T* o = out + (int64_t)m * 3 * HW + p;                 // [map][channel][pixel]; x / 255 * 2 - 1
        o[0] = (T)((float)c0 / 255.0f * 2.0f - 1.0f);
        o[HW] = (T)((float)c1 / 255.0f * 2.0f - 1.0f);
        o[2 * HW] = (T)((float)c2 / 255.0f * 2.0f - 1.0f);
    }
}

}  // namespace

extern "C" int pt_rasterize_tracks(const int32_t* pts, int32_t n_tracks, int32_t n_points, int32_t start, int32_t n_maps,
                                   int32_t n_total, int32_t H, int32_t W, int32_t flip_mode, int32_t out_is_f32, void* out,
                                   void* stream) {
    PT_CHECK(pts && out, "pt_rasterize_tracks: null pointer");
    PT_CHECK(n_tracks >= 0 && n_points >= 2 && start >= 0 && n_maps >= 0 && start + n_maps + 1 <= n_points + (n_maps == 0),
             "pt_rasterize_tracks: maps %d..%d need points up to %d of %d", start, start + n_maps, start + n_maps, n_points);
    PT_CHECK(n_total >= n_maps && n_total >= 1 && n_total < 65536 && H > 0 && W > 0 && (flip_mode == 0 || flip_mode == 1),
             "pt_rasterize_tracks: bad geometry");
    int64_t blocks = ((int64_t)H * W + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (out_is_f32)
        hipLaunchKernelGGL(rasterize_tracks_kernel<float>, dim3((unsigned)blocks, (unsigned)n_total), dim3(256), 0, (hipStream_t)stream,
                           pts, n_tracks, n_points, start, n_maps, n_total, H, W, flip_mode, (float*)out);
    else
        hipLaunchKernelGGL(rasterize_tracks_kernel<f16>, dim3((unsigned)blocks, (unsigned)n_total), dim3(256), 0, (hipStream_t)stream,
                           pts, n_tracks, n_points, start, n_maps, n_total, H, W, flip_mode, (f16*)out);
    PT_LAUNCH_CHECK("pt_rasterize_tracks");
    return 0;
}
