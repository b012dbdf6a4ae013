// Shared device helpers for the gfx950 ego-prior kernels.
// Everything here is compiled with -ffp-contract=off: the hit test, the pixel-centre formula and the projection
// must round exactly like the separate multiply/add sequences they restate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "egoprior.h"

#define EP_TILE EGOPRIOR_TILE
#define EP_K EGOPRIOR_MAX_K
#define EP_WAVE 64

namespace ep {

// pytorch3d PixToNonSquareNdc (rasterization_utils): NDC coordinate of pixel-centre i on an axis of S1 pixels whose
// other axis has S2 pixels.
__device__ __forceinline__ float pix_to_ndc(int i, int S1, int S2) {
    float range = 2.0f;
    if (S1 > S2) range = ((float)S1 * range) / (float)S2;
    const float offset = range / 2.0f;
    return -offset + (range * (float)i + offset) / (float)S1;
}

// Geometry of the raster shared by the binning and splat kernels (filled on the host).
struct RasterGeom {
    int H, W;
    int tiles_x, tiles_y;
    float half_range_x, half_range_y;  // NDC half extent of each axis
    float inv_pitch_x, inv_pitch_y;    // pixels per NDC unit
    float rad_px_x, rad_px_y;          // radius in pixels + conservative margin
    float radius2;                     // float32(r) * float32(r)
    int span_x, span_y;                // most tiles a disc's rectangle takes on each axis
};

// Tile rectangle [tx0,tx1] x [ty0,ty1] of the pixels whose centres can lie inside a disc; false when there is none (the
// disc misses the image or falls between pixel centres; also for non-finite coordinates and z < 0, which pytorch3d's
// rasterizer skips).  cx, cy are the disc centre in pixel-index units (pixel i has its centre at i), rad_px carries a
// margin far above the rounding of that conversion, so the candidates are the pixel indices in [ceil(lo), floor(hi)].
__device__ __forceinline__ bool tile_range(const RasterGeom& g, float px, float py, float pz, int& tx0, int& tx1,
                                           int& ty0, int& ty1) {
    // Straight-line code: every lane computes the rectangle and the tests only decide whether it counts.  The clamps
    // (median of three with 0 and W-1) change nothing for a disc that passes the tests and keep the conversions of one
    // that fails (infinite or NaN coordinates included) inside the integer range.
    const float wmax = (float)(g.W - 1), hmax = (float)(g.H - 1);
    const float cx = (g.half_range_x - px) * g.inv_pitch_x - 0.5f;
    const float cy = (g.half_range_y - py) * g.inv_pitch_y - 0.5f;
    const float lox = cx - g.rad_px_x, hix = cx + g.rad_px_x;
    const float loy = cy - g.rad_px_y, hiy = cy + g.rad_px_y;
    const bool inside = (pz >= 0.0f) & (hix >= 0.0f) & (lox <= wmax) & (hiy >= 0.0f) & (loy <= hmax);
    const int x0 = (int)ceilf(__builtin_amdgcn_fmed3f(lox, 0.0f, wmax));
    const int x1 = (int)floorf(__builtin_amdgcn_fmed3f(hix, 0.0f, wmax));
    const int y0 = (int)ceilf(__builtin_amdgcn_fmed3f(loy, 0.0f, hmax));
    const int y1 = (int)floorf(__builtin_amdgcn_fmed3f(hiy, 0.0f, hmax));
    tx0 = (int)((uint32_t)x0 / EP_TILE);
    tx1 = (int)((uint32_t)x1 / EP_TILE);
    ty0 = (int)((uint32_t)y0 / EP_TILE);
    ty1 = (int)((uint32_t)y1 / EP_TILE);
    return inside & (x0 <= x1) & (y0 <= y1);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (EP_WAVE - 1)); }

}  // namespace ep
