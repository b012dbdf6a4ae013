/*
 * egoprior.h -- C ABI of the MI355X (gfx950) ego-prior renderer.
 *
 * Every entry point takes plain pointers and sizes.  Pointers named d_* are DEVICE pointers owned by the caller
 * (inputs const, outputs and workspace pre-allocated by the caller); `stream` is a hipStream_t passed as void*.
 * All calls are asynchronous on `stream`, allocate nothing, and return 0 on success or a negative EGOPRIOR_E_* code;
 * egoprior_last_error() then returns a message (thread-local).  No entry point touches the host copy of any device
 * buffer; status words that kernels produce (counts, overflow flags) live in caller-provided device memory.
 *
 * The reference this replaces is scripts/render_vipe_pointcloud.py of kdh8156/EgoX-EgoPriorRenderer; each entry point
 * cites the reference lines it stands in for (paths relative to the reference checkout).  Where the reference delegates
 * to pytorch3d v0.7.9 the pytorch3d function is named too.
 */
#ifndef EGOPRIOR_H_
#define EGOPRIOR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGOPRIOR_ABI_VERSION 2
#define EGOPRIOR_MAX_K 20 /* points_per_pixel the reference always uses (:861, :1073) */
#define EGOPRIOR_TILE 8   /* binning tile edge in pixels: one 64-lane wave owns one 8x8 tile */

enum {
    EGOPRIOR_OK = 0,
    EGOPRIOR_E_INVALID = -1,   /* bad argument (null pointer, non-positive size, unknown enum) */
    EGOPRIOR_E_WORKSPACE = -2, /* workspace too small for the request */
    EGOPRIOR_E_HIP = -3,       /* a HIP runtime call or kernel launch failed */
    EGOPRIOR_E_UNSUPPORTED = -4
};

/* which pixels of a frame become points (instance-mask rule) */
enum {
    EGOPRIOR_SELECT_STATIC = 0,  /* mask == 0  (all pixels when d_mask is NULL)  -- :409-416, :547-552 */
    EGOPRIOR_SELECT_DYNAMIC = 1, /* mask != 0  (no pixel when d_mask is NULL)    -- :725-734 */
    EGOPRIOR_SELECT_ALL = 2      /* ignore the mask                              -- :635 (mean cloud) */
};

enum { EGOPRIOR_CAM_PINHOLE = 0, EGOPRIOR_CAM_FISHEYE = 1 };

/* One point of a device-resident cloud: world xyz (float32) + colour bytes r | g<<8 | b<<16 in `rgba`. */
typedef struct egoprior_point {
    float x, y, z;
    uint32_t rgba;
} egoprior_point;

/* Exo-frame unprojection parameters (one ViPE frame). */
typedef struct egoprior_unproject_params {
    int32_t H, W;      /* full-resolution frame size */
    int32_t subsample; /* spatial_subsample, >= 1 (reference default 2) */
    int32_t select;    /* EGOPRIOR_SELECT_* */
    float fx, fy, cx, cy; /* ViPE pinhole intrinsics of the frame */
    float c2w[12];     /* exo camera -> ViPE world, row-major 3x4, float32 (pose npz `data`) */
    double T[12];      /* ViPE world -> Ego4D world (= inv(exo W2C) promoted to float64), row-major 3x4 */
} egoprior_unproject_params;

/* Ego camera in pytorch3d's row-vector convention, Xv = Xw * R + T.  The host fills it exactly as the reference's
 * host code does (cameras_from_opencv_projection for pinhole, :990-1044 for fisheye). */
typedef struct egoprior_camera {
    int32_t model; /* EGOPRIOR_CAM_* */
    int32_t H, W;  /* raster size (rows, columns) */
    float R[9];    /* row-major 3x3 */
    float T[3];
    float fx, fy;  /* NDC focal length (fisheye: fx == fy == f_ndc) */
    float px, py;  /* NDC principal point */
    float radial[6], tangential[2], thin_prism[4]; /* fisheye only */
} egoprior_camera;

typedef struct egoprior_raster_params {
    float radius;        /* float32(radius_ndc), radius_ndc = clamp(point_size*2/max(W,H), 1e-5, 0.25)  (:853-855) */
    float radius_sq_w;   /* float32(radius_ndc*radius_ndc) with the product taken in double: the weights' denominator */
    int32_t K;           /* points_per_pixel, 1..EGOPRIOR_MAX_K */
    int32_t rotate_cw;   /* 1: write the uint8 frame rotated 90 deg clockwise (W rows x H columns) (:888) */
    int32_t colors_gt1;  /* 1: colour bytes are divided by 255 (the reference's `if cols.max() > 1` rule, :833-834);
                            0: used as they are; -1: decide from the per-cloud flags d_flags_a/d_flags_b */
} egoprior_raster_params;

const char* egoprior_last_error(void);
int egoprior_abi_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * reliable_depth_mask_range(depth)  --  vipe/utils/depth.py:385-422.
 * d_depth (H,W) float32 -> d_mask_out (H,W) uint8 (1 = reliable).
 * --------------------------------------------------------------------------------------------------------------- */
int egoprior_reliable_depth_mask(const float* d_depth, int32_t H, int32_t W, uint8_t* d_mask_out, void* stream);
/* The *_f16 variants below take the depth map as IEEE float16 bit patterns, the precision the depth artifacts are
 * stored in (EXR channel Z, pixel type HALF: vipe/utils/io.py:274-312, which widens them to float32 on the host).
 * Widening on the device is exact, so the results are the same bits; the upload and the HBM footprint halve. */
int egoprior_reliable_depth_mask_f16(const uint16_t* d_depth, int32_t H, int32_t W, uint8_t* d_mask_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * One exo frame -> compacted points.  Stands in for the per-frame body of build_background_pointcloud (:358-430,
 * 442-445), build_dynamic_points_for_frame (:682-751) and the unprojection half of
 * build_mean_background_pointcloud (:594-654): rays * depth, c2w, reliability mask & instance rule on the [::s, ::s]
 * grid, row-major compaction, exo->world in float64.
 *   d_depth (H,W) f32, d_rgb (H,W,3) u8, d_mask (H,W) u8 or NULL.
 *   d_points: capacity ceil(H/s)*ceil(W/s) records; d_xyz64 (optional, same capacity x 3 doubles) receives the
 *   float64 coordinates the reference returns to Python; d_count: one int32, number of points written;
 *   d_flags: one int32, set to 1 when any written colour byte exceeds 1 (else 0).
 * Workspace: egoprior_unproject_workspace_bytes(H, W, subsample).
 * --------------------------------------------------------------------------------------------------------------- */
size_t egoprior_unproject_workspace_bytes(int32_t H, int32_t W, int32_t subsample);
int egoprior_unproject_frame(const egoprior_unproject_params* p, const float* d_depth, const uint8_t* d_rgb,
                             const uint8_t* d_mask, egoprior_point* d_points, double* d_xyz64, int32_t* d_count,
                             int32_t* d_flags, void* d_workspace, size_t workspace_bytes, void* stream);
int egoprior_unproject_frame_f16(const egoprior_unproject_params* p, const uint16_t* d_depth, const uint8_t* d_rgb,
                                 const uint8_t* d_mask, egoprior_point* d_points, double* d_xyz64, int32_t* d_count,
                                 int32_t* d_flags, void* d_workspace, size_t workspace_bytes, void* stream);
/* The same with a ray table instead of the pinhole formula: d_rays (ceil(H/s), ceil(W/s), 3) float32, one ray per
 * sample, as scripts/render_vipe_pointcloud.py:384-398 builds it from CameraType.build_camera_model(intr).iproj_disp
 * for the MEI (vipe/utils/cameras.py:228-250) and PANORAMA (:366-387) exo cameras; points are rays * depth.  fx, fy,
 * cx, cy of `p` are ignored.  depth_is_f16 selects the depth type (0: float32, 1: float16 bit patterns). */
int egoprior_unproject_frame_rays(const egoprior_unproject_params* p, const void* d_depth, int32_t depth_is_f16,
                                  const float* d_rays, const uint8_t* d_rgb, const uint8_t* d_mask,
                                  egoprior_point* d_points, double* d_xyz64, int32_t* d_count, int32_t* d_flags,
                                  void* d_workspace, size_t workspace_bytes, void* stream);

/* egoprior_unproject_frame for callers that cannot allocate device memory or carry a workspace: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same inputs, outputs
 * and results as egoprior_unproject_frame (float32 depth); the selection flags and block counters of the compaction
 * live in a device array that is part of the library image (about 8.1 MiB per GPU the library runs on, held until the
 * process ends) instead of a caller-owned workspace.  It calls no hipMalloc / hipFree or allocating routine.
 * Limits: ceil(H/s) * ceil(W/s) <= EGOPRIOR_NOALLOC_MAX_SAMPLES (a 3840 x 2160 frame at subsample 1 fits), otherwise
 * EGOPRIOR_E_UNSUPPORTED; every call on a GPU shares that GPU's array, so calls that overlap in time on one GPU
 * (different streams, different threads) race -- serialise them, or use egoprior_unproject_frame with one workspace
 * per stream.  The first call on each device resolves the array's address (hipGetSymbolAddress): make that call outside
 * graph capture.  The device that owns `stream` must be the current device (each GPU has its own copy of the array, at
 * its own address); a stream of another device is refused with EGOPRIOR_E_INVALID. */
#define EGOPRIOR_NOALLOC_MAX_SAMPLES (8 * 1024 * 1024 + 256 * 1024)
int egoprior_unproject_frame_noalloc(const egoprior_unproject_params* p, const float* d_depth, const uint8_t* d_rgb,
                                     const uint8_t* d_mask, egoprior_point* d_points, double* d_xyz64,
                                     int32_t* d_count, int32_t* d_flags, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Mean background  --  compute_robust_mean_tensors (:459-473) + the frame loop of build_mean_background_pointcloud
 * (:522-583) + colour quantisation (:638).
 * State per pixel: 4 cascade levels x (depth, r, g, b) float32 sums + int32 count, planar, 17*H*W*4 bytes, zeroed by
 * the caller before the first accumulate call.  Frames are d_depth (F,H,W) f32, d_rgb (F,H,W,3) u8, d_mask (F,H,W) u8
 * or NULL; frame f of this call is term number first_index+f of the per-pixel sums (the order of summation follows
 * torch.nanmean on CPU: runs of 16 terms, 4 levels).
 * finalize: mean_depth (H,W) f32 with NaN->0, mean colour bytes (H,W,3) = trunc(clamp(mean*255, 0, 255)).
 * egoprior_meanbg_export / _import move the order-insensitive form of a state (folded sums + count, 5 planes) so that
 * ranks can all-reduce it; after an import the state holds the folded sums at level 0.
 * --------------------------------------------------------------------------------------------------------------- */
size_t egoprior_meanbg_state_bytes(int32_t H, int32_t W);
int egoprior_meanbg_accumulate(float* d_state, const float* d_depth, const uint8_t* d_rgb, const uint8_t* d_mask,
                               int32_t F, int64_t first_index, int32_t H, int32_t W, void* stream);
int egoprior_meanbg_accumulate_f16(float* d_state, const uint16_t* d_depth, const uint8_t* d_rgb, const uint8_t* d_mask,
                               int32_t F, int64_t first_index, int32_t H, int32_t W, void* stream);
int egoprior_meanbg_finalize(const float* d_state, int32_t H, int32_t W, float* d_mean_depth, uint8_t* d_mean_rgb,
                             void* stream);
/* The viewer's mean background, compute_mean_background_depth + compute_robust_mean_tensors of vipe/utils/viser.py:60-74,
 * 123-214: the frames are masked at full resolution and subsampled [::s, ::s] before they are stacked, and the means stay
 * float32 (no colour quantisation).  From the same state as egoprior_meanbg_finalize:
 *   d_mean_depth (ceil(H/s), ceil(W/s)) f32, d_mean_rgb (ceil(H/s), ceil(W/s), 3) f32, NaN (pixels no frame kept) -> 0. */
int egoprior_meanbg_finalize_f32(const float* d_state, int32_t H, int32_t W, int32_t subsample, float* d_mean_depth,
                                 float* d_mean_rgb, void* stream);
int egoprior_meanbg_export(const float* d_state, int32_t H, int32_t W, float* d_sums4, int32_t* d_count, void* stream);
int egoprior_meanbg_import(float* d_state, int32_t H, int32_t W, const float* d_sums4, const int32_t* d_count,
                           void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Render-entry filters of render_points_pytorch3d / render_points_fisheye (:800-824, :953-974): float32 cast is the
 * caller's, this drops rows with a non-finite coordinate or |p| >= 1e6 and compacts in order.
 *   d_xyz (N,3) f32, d_rgb (N,3) u8 -> d_points (capacity N), d_count, d_flags as in egoprior_unproject_frame.
 * Workspace: egoprior_filter_workspace_bytes(N).
 * --------------------------------------------------------------------------------------------------------------- */
size_t egoprior_filter_workspace_bytes(int64_t N);
int egoprior_filter_cloud(const float* d_xyz, const uint8_t* d_rgb, int64_t N, egoprior_point* d_points,
                          int32_t* d_count, int32_t* d_flags, void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * PointsRasterizer.transform (pytorch3d) as used at :863-871 / :1075-1083: world -> view -> NDC x,y with z = view
 * depth.  Cloud = [a | b] (b may be NULL); the live length of each part is n_* unless d_n_* (device int32) is given,
 * in which case n_* is the capacity.  d_ndc receives (x_ndc, y_ndc, z_view, rgba) per point.
 * --------------------------------------------------------------------------------------------------------------- */
int egoprior_project_points(const egoprior_camera* cam, const egoprior_point* d_a, int64_t n_a, const int32_t* d_n_a,
                            const egoprior_point* d_b, int64_t n_b, const int32_t* d_n_b, egoprior_point* d_ndc,
                            void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * pytorch3d rasterize_points(points_ndc, image_size=(H,W), radius, points_per_pixel=K) + PointsRenderer weights +
 * AlphaCompositor(background (0,0,0)) + the reference's post-processing (:874-888, :1086-1097), on points already in
 * NDC (d_ndc from egoprior_project_points, n = capacity, d_n = optional live count).
 * Outputs (each optional, NULL to skip):
 *   d_image_u8  (H,W,3) uint8, or (W,H,3) when rotate_cw      -- the frame the reference returns
 *   d_image_f32 (H,W,3) float32                               -- compositor output before clamp / dark rule
 *   d_idx, d_zbuf, d_dists (H,W,K) int32 / f32 / f32, -1 pad  -- rasterize_points' three outputs
 * Workspace: egoprior_raster_workspace_bytes(H, W, n, radius); a frame whose tile lists outgrow it sets the int32 at
 * d_status to 1 and drops the entries that do not fit.  The word is never cleared by the library (zero it before the
 * first call), so one word can watch over many frames.  The size the query returns reserves, for every point,
 * the largest number of tiles a disc of that radius can overlap (8 bytes per point-tile pair), so it cannot overflow
 * for that n.  Any smaller workspace is accepted too (at least egoprior_raster_workspace_bytes_entries(H, W, n, 0)):
 * the lists then hold what fits and d_status reports the rest; egoprior_count_list_entries gives the exact need of a
 * frame beforehand.  About a ninth of the size is the slice area: a tile whose list exceeds 32768 entries is cut into
 * slices that separate waves reduce to per-pixel lists (10 KiB per slice) before one of them merges the lists and
 * shades the tile; lists and slice area share the workspace in a fixed ratio, whatever its size.  List positions are 32-bit: a call with n times that number of tiles above 2^32 - 16 is refused
 * with EGOPRIOR_E_INVALID (split the cloud).
 * --------------------------------------------------------------------------------------------------------------- */
size_t egoprior_raster_workspace_bytes(int32_t H, int32_t W, int64_t n, float radius);
size_t egoprior_raster_workspace_bytes_entries(int32_t H, int32_t W, int64_t n, int64_t entries);
int egoprior_rasterize_composite(const egoprior_raster_params* rp, int32_t H, int32_t W, const egoprior_point* d_ndc,
                                 int64_t n, const int32_t* d_n, const int32_t* d_flags_a, const int32_t* d_flags_b,
                                 uint8_t* d_image_u8, float* d_image_f32, int32_t* d_idx, float* d_zbuf,
                                 float* d_dists, int32_t* d_status, void* d_workspace, size_t workspace_bytes,
                                 void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * One ego frame end to end = render_points_pytorch3d (:754-890) or render_points_fisheye (:892-1102):
 * egoprior_project_points followed by egoprior_rasterize_composite, sharing one workspace.  The reference's entry
 * filters (:807-821: finite coordinates, float32 norm below 1e6) are applied to both parts on the fly: a point that fails
 * them is never drawn but keeps its place in the index space (run egoprior_filter_cloud first where the index maps must
 * count only surviving points, as the reference's do).
 * Workspace: egoprior_render_workspace_bytes(H, W, n_a + n_b, radius).
 * --------------------------------------------------------------------------------------------------------------- */
size_t egoprior_render_workspace_bytes(int32_t H, int32_t W, int64_t n, float radius);
size_t egoprior_render_workspace_bytes_entries(int32_t H, int32_t W, int64_t n, int64_t entries);
int egoprior_render_frame(const egoprior_camera* cam, const egoprior_raster_params* rp, const egoprior_point* d_a,
                          int64_t n_a, const int32_t* d_n_a, const int32_t* d_flags_a, const egoprior_point* d_b,
                          int64_t n_b, const int32_t* d_n_b, const int32_t* d_flags_b, uint8_t* d_image_u8,
                          float* d_image_f32, int32_t* d_idx, float* d_zbuf, float* d_dists, int32_t* d_status,
                          void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * The renderers' post-processing with a background image that is not black (:794-797 / :947-950 build it with
 * generate_background, :873-888 / :1085-1097 apply it): d_image_f32 is the (H,W,3) compositor output of
 * egoprior_rasterize_composite / egoprior_render_frame, d_background_f32 an (H,W,3) float32 image in 0..1.  A pixel whose
 * clamped channels are all below 1e-2 takes the background's pixel; the result is clipped, scaled by 255 and truncated
 * to uint8, and rotated by 90 degrees clockwise into a (W,H,3) image when rotate_cw is set.  (With a black background
 * this is what those two entry points write to d_image_u8 themselves.)
 * --------------------------------------------------------------------------------------------------------------- */
int egoprior_postprocess_background(const float* d_image_f32, const float* d_background_f32, int32_t H, int32_t W,
                                    int32_t rotate_cw, uint8_t* d_image_u8, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Exact sizing of the tile lists (optional first call of a two-call mode; no counterpart in the reference, whose
 * pytorch3d rasterizer caps its bins and drops points instead).  Runs the projection (cam != NULL: [a | b] are world
 * points; cam == NULL: d_a holds n_a NDC records, image size H x W, no part b) and the per-tile counting pass, and
 * writes the number of list entries the frame needs to the device word d_entries.  After reading it back, a workspace
 * of egoprior_raster_workspace_bytes_entries / egoprior_render_workspace_bytes_entries (H, W, n, entries) holds that
 * frame; discs of 40 pixels on a large image would otherwise reserve dozens of tiles per point.
 * Workspace of this call: egoprior_raster_workspace_bytes_entries(H, W, n, 0).
 * --------------------------------------------------------------------------------------------------------------- */
int egoprior_count_list_entries(const egoprior_camera* cam, const egoprior_raster_params* rp, int32_t H, int32_t W,
                                const egoprior_point* d_a, int64_t n_a, const int32_t* d_n_a,
                                const egoprior_point* d_b, int64_t n_b, const int32_t* d_n_b, int64_t* d_entries,
                                void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * White-pixel metric of finished frames, data_preprocess/scripts/postprocess_render_dir.py:17-31 (grey conversion and
 * threshold) and :46-67 (per-frame counts that the batch script sums up to pick the best camera), computed on the
 * uint8 frames where the renderer left them instead of on PNG files dumped by ffmpeg.
 * d_frames: n_frames x H x W x 3 uint8; d_counts[n_frames] receives the number of pixels whose grey value reaches
 * threshold; d_masks (optional, n_frames x H x W) receives 0 / 255.  grey_rule picks the script's grey conversion:
 * EGOPRIOR_GREY_PIL_L, its first branch (:17-19, taken whenever PIL imports; pillow is pinned in the reference's
 * envs/requirements.txt): PIL's Image.convert("L") = (19595 r + 38470 g + 7471 b + 0x8000) >> 16;
 * EGOPRIOR_GREY_FLOAT, its imageio fallback (:20-25): float32(0.299 r + 0.587 g + 0.114 b).
 * --------------------------------------------------------------------------------------------------------------- */
enum { EGOPRIOR_GREY_PIL_L = 0, EGOPRIOR_GREY_FLOAT = 1 };
int egoprior_white_pixel_counts(const uint8_t* d_frames, int32_t n_frames, int32_t H, int32_t W, float threshold,
                                int32_t grey_rule, int32_t* d_counts, uint8_t* d_masks, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Per-kernel timing for benchmarks (no counterpart in the reference, which has no profiling on this path).
 * Between _begin and _end every kernel launched by the calls above on the calling thread is bracketed by a pair of
 * HIP events on its launch stream.  _end waits for them, sums the elapsed time per kernel name and returns, for up to
 * max_kernels distinct kernels, newline-separated names, total milliseconds and launch counts.  _end synchronises
 * with the device: keep it out of timed regions.
 * --------------------------------------------------------------------------------------------------------------- */
int egoprior_profile_begin(void);
/* Process-wide tuning / test switches (no counterpart in the reference; the one piece of state the library keeps
 * besides the per-thread error text).  Every entry point reads them once when it starts, so a change takes effect with
 * the next call and never in the middle of one.  "splat_variant": 0 default sweep kernel (448-record chunks, three
 * waves per SIMD), 1 plain register-insertion kernel, 2 sweep kernel with 64-record chunks (forces its multi-chunk path
 * on small inputs), 3 with 1024-record chunks, 4 with 384-record chunks and four waves per SIMD, 5 with 512-record
 * chunks; every variant returns the same bits.
 * "fill_pieces": 1 default: the tile lists are filled without global atomics where the counting pass could leave its
 * per-workgroup offsets, 0: always one reservation per run of lanes.  "bin_lds": 1 default: the tile counts are
 * aggregated per workgroup in LDS, 0: straight global atomics.  "slices": 1 default: tiles with more than 32768 list
 * entries are split over several waves, 0: one wave per tile always.  "scan_threads": 256 default, or 1024: workgroup
 * size of the single-workgroup tile scan (256 threads find room on a compute unit that is full of splat waves; 1024 are
 * 3 us faster on an idle GPU).  "scan_once": 1 default: grids of up to 8192 tiles are scanned from registers after one
 * round of loads, 0: always the looping scan.  "tail_slots": 0 default; > 0 = the number of splat waves the GPU holds at once (3072 on
 * MI355X): when the tiles with real work would leave the last round of a launch mostly empty, the tiles of that round
 * are cut into three slices each (shortens a lone launch for some views). */
int egoprior_set_option(const char* name, int32_t value);
/* current value of an option, -1 for a name the library does not know */
int32_t egoprior_get_option(const char* name);
int egoprior_profile_end(char* names, size_t names_bytes, float* total_ms, int32_t* calls, int32_t max_kernels,
                         int32_t* n_kernels);

#ifdef __cplusplus
}
#endif
#endif /* EGOPRIOR_H_ */
