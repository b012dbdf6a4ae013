// Output-side helper for gfx950: the white-pixel metric the batch scripts compute on the dumped frames
// (data_preprocess/scripts/postprocess_render_dir.py:17-31, 46-67), evaluated on the frames while they are still in HBM.
//   gray = PIL's Image.convert("L"): (19595 r + 38470 g + 7471 b + 0x8000) >> 16, an integer in 0..255  (:17-19, the branch
//          the script takes whenever PIL imports, and pillow is pinned in the reference's envs/requirements.txt)
//       or float32(0.299 * r + 0.587 * g + 0.114 * b)   (:20-25, the imageio fallback; float64 products and sums, left to
//          right, as numpy does)
//   mask = float32(gray) >= threshold ;  per frame: number of mask pixels
#include "egoprior_common.h"
#include "egoprior_host.h"

namespace ep {

// One workgroup takes 1024 consecutive pixels of one frame (4 per thread); counts go to counts[frame] with one atomic
// per workgroup.  HBM-bound: 3 bytes in per pixel, 1 byte out when the mask image is asked for.
__global__ void __launch_bounds__(256) k_white_pixels(const uint8_t* __restrict__ frames, int64_t pixels_per_frame,
                                                      float threshold, int grey_rule,
                                                      int32_t* __restrict__ counts, uint8_t* __restrict__ masks) {
    __shared__ uint32_t ws[4];
    const int64_t f = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * 1024 + (int64_t)threadIdx.x * 4;
    const uint8_t* src = frames + f * pixels_per_frame * 3;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t p = p0 + k;
        if (p < pixels_per_frame) {
            const uint32_t ri = src[p * 3 + 0], gi = src[p * 3 + 1], bi = src[p * 3 + 2];
            float gray;
            if (grey_rule == EGOPRIOR_GREY_PIL_L) {
                gray = (float)((ri * 19595u + gi * 38470u + bi * 7471u + 0x8000u) >> 16);
            } else {
                const double r = (double)ri, g = (double)gi, b = (double)bi;
                gray = (float)((0.299 * r + 0.587 * g) + 0.114 * b);
            }
            const bool on = gray >= threshold;
            c += on ? 1u : 0u;
            if (masks != nullptr) masks[f * pixels_per_frame + p] = on ? 255 : 0;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = ws[0] + ws[1] + ws[2] + ws[3];
        if (t != 0u) atomicAdd(&counts[f], (int32_t)t);
    }
}

// The reference's post-processing with a background image (scripts/render_vipe_pointcloud.py:873-888, 1085-1097):
// rendered = clamp(compositor output, 0, 1); a pixel whose three channels are all below 1e-2 takes the background
// image's pixel instead; uint8 by truncation of clip(., 0, 1) * 255; optional 90-degree clockwise rotation
// (out[i][j] = in[H-1-j][i], out is W x H).
__global__ void __launch_bounds__(256) k_postprocess_background(const float* __restrict__ img,
                                                                const float* __restrict__ bg, int H, int W,
                                                                int rotate_cw, uint8_t* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
    float c[3];
    bool dark = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c[k] = fminf(fmaxf(img[p * 3 + k], 0.0f), 1.0f);
        dark = dark && (c[k] < 1e-2f);
    }
    const size_t o = rotate_cw ? ((size_t)x * H + (size_t)(H - 1 - y)) : (size_t)p;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = dark ? bg[p * 3 + k] : c[k];
        out[o * 3 + k] = (uint8_t)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f);
    }
}

}  // namespace ep

extern "C" int egoprior_postprocess_background(const float* d_image_f32, const float* d_background_f32, int32_t H,
                                               int32_t W, int32_t rotate_cw, uint8_t* d_image_u8, void* stream) {
    if (d_image_f32 == nullptr || d_background_f32 == nullptr || d_image_u8 == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (H <= 0 || W <= 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad sizes");
    const int64_t n = (int64_t)H * W;
    EP_LAUNCH("k_postprocess_background", ep::k_postprocess_background, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
              (hipStream_t)stream, d_image_f32, d_background_f32, (int)H, (int)W, (int)rotate_cw, d_image_u8);
    return ep_host::check_launch("postprocess_background");
}

extern "C" int egoprior_white_pixel_counts(const uint8_t* d_frames, int32_t n_frames, int32_t H, int32_t W,
                                           float threshold, int32_t grey_rule, int32_t* d_counts, uint8_t* d_masks,
                                           void* stream) {
    if (d_frames == nullptr || d_counts == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (n_frames < 0 || H <= 0 || W <= 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad sizes");
    if (grey_rule != EGOPRIOR_GREY_PIL_L && grey_rule != EGOPRIOR_GREY_FLOAT)
        return ep_host::fail(EGOPRIOR_E_INVALID, "unknown grey rule");
    if (n_frames == 0) return 0;
    if (n_frames > 65535) return ep_host::fail(EGOPRIOR_E_UNSUPPORTED, "more than 65535 frames per call");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)n_frames, s) != hipSuccess)
        return ep_host::fail(EGOPRIOR_E_HIP, "memset failed");
    const int64_t ppf = (int64_t)H * W;
    EP_LAUNCH("k_white_pixels", ep::k_white_pixels, dim3((unsigned)((ppf + 1023) / 1024), (unsigned)n_frames), dim3(256),
              0, s, d_frames, ppf, threshold, (int)grey_rule, d_counts, d_masks);
    return ep_host::check_launch("white_pixel_counts");
}
