/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in the product path links,
 * loads or calls this file; it exists to check the HIP kernels.
 *
 * Plain-C restatement of the arithmetic that the reference delegates to
 * pytorch3d v0.7.9 (third-party, pinned in reference README.md:16, not present
 * in /root/reference and not installable here).  The call sites it restates:
 *   scripts/render_vipe_pointcloud.py:858-871   (pinhole: settings + render)
 *   scripts/render_vipe_pointcloud.py:1070-1083 (fisheye: settings + render)
 * and, inside pytorch3d (restated from its published v0.7.x source):
 *   csrc/rasterize_points/rasterize_points_cpu.cpp  RasterizePointsNaiveCpu
 *   csrc/rasterize_points/rasterization_utils.h     PixToNonSquareNdc
 *   csrc/compositing/alpha_composite_cpu.cpp        alphaCompositeCpuForward
 *   renderer/points/renderer.py                     weights = 1 - dists2/(r*r)
 *   renderer/points/compositor.py                   background fill where idx[...,0] < 0
 *
 * PARITY UNPINNED for this file: the reference ships no golden vectors, tests
 * or fixtures for this boundary and pytorch3d cannot be run here, so these
 * functions follow the published algorithm and are pinned only by hand-derived
 * known-answer cases (tests/test_oracle_known_answers.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * FMA contraction must stay off: the hit test d2 < r2 decides index maps.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* pytorch3d rasterization_utils: pixel index -> NDC coordinate of the pixel
 * centre for a possibly non-square image.  S1 is the size of the axis being
 * converted, S2 the other axis.  Float semantics (range = S1*2/S2). */
static inline float pix_to_non_square_ndc(int i, int S1, int S2) {
    float range = 2.0f;
    if (S1 > S2) {
        range = ((float)S1 * range) / (float)S2;
    }
    const float offset = range / 2.0f;
    return -offset + (range * (float)i + offset) / (float)S1;
}

float oracle_pix_to_ndc(int i, int S1, int S2) { return pix_to_non_square_ndc(i, S1, S2); }

typedef struct {
    float z;
    int32_t idx;
    float d2;
} frag_t;

/* strict "a before b" in the (z, idx) order std::tuple<float,int,float> gives
 * (d2 can never decide: idx is unique) */
static inline int frag_less(float za, int32_t ia, float zb, int32_t ib) {
    if (za < zb) return 1;
    if (zb < za) return 0;
    return ia < ib;
}

/* Insert into a list kept sorted ascending by (z, idx), keeping at most K.
 * Same retained set and output order as pytorch3d's max-heap that pops when
 * size > K and then drains back-to-front. */
static inline void frag_insert(frag_t* q, int* n, int K, float z, int32_t idx, float d2) {
    int c = *n;
    if (c == K) {
        if (!frag_less(z, idx, q[K - 1].z, q[K - 1].idx)) return;
        c = K - 1;
    }
    int i = c;
    while (i > 0 && frag_less(z, idx, q[i - 1].z, q[i - 1].idx)) {
        q[i] = q[i - 1];
        --i;
    }
    q[i].z = z;
    q[i].idx = idx;
    q[i].d2 = d2;
    *n = c + 1;
}

/*
 * RasterizePointsNaiveCpu for one cloud.
 *   pts   : (P,3) float32 NDC x, NDC y, view depth z
 *   radius: (P,)  float32 per-point NDC radius
 *   rows [y0, y1) only (lets a caller time or check a band of the image);
 *   outputs are full (H,W,K) arrays, only rows in the band are written.
 * Unfilled slots = -1 (idx, zbuf, dists).
 */
void oracle_rasterize_points_naive(const float* pts, const float* radius, int64_t P, int H, int W, int K,
                                   int y0, int y1, int32_t* idx_out, float* zbuf_out, float* dist_out) {
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
#pragma omp parallel
    {
        frag_t* q = (frag_t*)malloc(sizeof(frag_t) * (size_t)K);
#pragma omp for schedule(dynamic, 1)
        for (int yi = y0; yi < y1; ++yi) {
            const int yidx = H - 1 - yi; /* +Y up */
            const float yf = pix_to_non_square_ndc(yidx, H, W);
            for (int xi = 0; xi < W; ++xi) {
                const int xidx = W - 1 - xi; /* +X left */
                const float xf = pix_to_non_square_ndc(xidx, W, H);
                int n = 0;
                for (int64_t p = 0; p < P; ++p) {
                    const float px = pts[3 * p + 0];
                    const float py = pts[3 * p + 1];
                    const float pz = pts[3 * p + 2];
                    const float pr = radius[p];
                    const float radius2 = pr * pr;
                    if (pz < 0) continue;
                    const float dx = px - xf;
                    const float dy = py - yf;
                    const float dist2 = dx * dx + dy * dy;
                    if (dist2 < radius2) {
                        frag_insert(q, &n, K, pz, (int32_t)p, dist2);
                    }
                }
                const size_t base = ((size_t)yi * (size_t)W + (size_t)xi) * (size_t)K;
                for (int k = 0; k < K; ++k) {
                    if (k < n) {
                        idx_out[base + k] = q[k].idx;
                        zbuf_out[base + k] = q[k].z;
                        dist_out[base + k] = q[k].d2;
                    } else {
                        idx_out[base + k] = -1;
                        zbuf_out[base + k] = -1.0f;
                        dist_out[base + k] = -1.0f;
                    }
                }
            }
        }
        free(q);
    }
}

/*
 * Same outputs as oracle_rasterize_points_naive, computed point-major: each
 * point visits only the pixels of a conservatively padded bounding box and the
 * identical hit test / insertion runs there.  The retained set is the K
 * smallest (z, idx) among hits, which does not depend on visiting order, so
 * the result is identical; tests/test_oracle_known_answers.py checks that on
 * random clouds.  Exists so full-size frames can be checked in seconds.
 * All points must share one radius r (what the reference always passes).
 */
void oracle_rasterize_points_scatter(const float* pts, float r, int64_t P, int H, int W, int K,
                                     int32_t* idx_out, float* zbuf_out, float* dist_out) {
    const size_t npix = (size_t)H * (size_t)W;
    frag_t* q = (frag_t*)malloc(sizeof(frag_t) * npix * (size_t)K);
    int* cnt = (int*)calloc(npix, sizeof(int));
    float* xf = (float*)malloc(sizeof(float) * (size_t)W);
    float* yf = (float*)malloc(sizeof(float) * (size_t)H);
    for (int xi = 0; xi < W; ++xi) xf[xi] = pix_to_non_square_ndc(W - 1 - xi, W, H);
    for (int yi = 0; yi < H; ++yi) yf[yi] = pix_to_non_square_ndc(H - 1 - yi, H, W);
    const float radius2 = r * r;
    /* NDC pitch of one pixel along each axis */
    const double range_x = (W > H) ? (2.0 * W) / H : 2.0;
    const double range_y = (H > W) ? (2.0 * H) / W : 2.0;
    const double pitch_x = range_x / W, pitch_y = range_y / H;
    /* Threads split the image into bands of rows; every thread walks all points in index order and touches only the
     * pixels of its own band, so each pixel still sees its points in the same order as in a serial run. */
#pragma omp parallel
    {
        int nt = 1, tid = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads();
        tid = omp_get_thread_num();
#endif
        const int band0 = (int)(((int64_t)H * tid) / nt), band1 = (int)(((int64_t)H * (tid + 1)) / nt) - 1;
        for (int64_t p = 0; p < P; ++p) {
            const float px = pts[3 * p + 0], py = pts[3 * p + 1], pz = pts[3 * p + 2];
            if (pz < 0) continue;
            if (!(isfinite(px) && isfinite(py))) continue; /* cannot pass dist2 < radius2 */
            /* xf[xi] = range_x/2 - (xi + 0.5) * pitch_x  =>  xi = (range_x/2 - x)/pitch_x - 0.5 */
            const double cx = (range_x / 2.0 - (double)px) / pitch_x - 0.5;
            const double cy = (range_y / 2.0 - (double)py) / pitch_y - 0.5;
            const double rx = (double)r / pitch_x + 1.0, ry = (double)r / pitch_y + 1.0; /* +1 px pad */
            double lo_x = floor(cx - rx), hi_x = ceil(cx + rx), lo_y = floor(cy - ry), hi_y = ceil(cy + ry);
            if (hi_x < 0 || hi_y < 0 || lo_x > W - 1 || lo_y > H - 1) continue;
            const int x0 = lo_x < 0 ? 0 : (int)lo_x, x1 = hi_x > W - 1 ? W - 1 : (int)hi_x;
            int y0 = lo_y < 0 ? 0 : (int)lo_y, y1 = hi_y > H - 1 ? H - 1 : (int)hi_y;
            if (y0 < band0) y0 = band0;
            if (y1 > band1) y1 = band1;
            for (int yi = y0; yi <= y1; ++yi) {
                const float dy = py - yf[yi];
                for (int xi = x0; xi <= x1; ++xi) {
                    const float dx = px - xf[xi];
                    const float dist2 = dx * dx + dy * dy;
                    if (dist2 < radius2) {
                        const size_t pix = (size_t)yi * (size_t)W + (size_t)xi;
                        frag_insert(q + pix * (size_t)K, &cnt[pix], K, pz, (int32_t)p, dist2);
                    }
                }
            }
        }
    }
    for (size_t pix = 0; pix < npix; ++pix) {
        const int n = cnt[pix];
        for (int k = 0; k < K; ++k) {
            const size_t o = pix * (size_t)K + (size_t)k;
            if (k < n) {
                idx_out[o] = q[o].idx;
                zbuf_out[o] = q[o].z;
                dist_out[o] = q[o].d2;
            } else {
                idx_out[o] = -1;
                zbuf_out[o] = -1.0f;
                dist_out[o] = -1.0f;
            }
        }
    }
    free(q);
    free(cnt);
    free(xf);
    free(yf);
}

/*
 * PointsRenderer weights + AlphaCompositor(background_color=(0,0,0)).
 *   idx, dists : (H,W,K) from the rasterizer
 *   feats      : (P,C) float32 colours
 *   inv_denom  : the float32 value of r*r computed in double (python float
 *                product cast to the tensor dtype)
 *   out        : (H,W,C) float32
 * Per pixel and channel: cum=1; for k: skip idx<0; a = 1 - d2/rr;
 * out += (cum*a)*f; cum = cum*(1-a).  Pixels whose first slot is empty get
 * the background colour.
 */
void oracle_alpha_composite(const int32_t* idx, const float* dists, const float* feats, int C, int H, int W,
                            int K, float rr, const float* background, float* out) {
    const size_t npix = (size_t)H * (size_t)W;
#pragma omp parallel for schedule(static)
    for (int64_t pix = 0; pix < (int64_t)npix; ++pix) {
        const int32_t* pi = idx + (size_t)pix * (size_t)K;
        const float* pd = dists + (size_t)pix * (size_t)K;
        for (int c = 0; c < C; ++c) {
            float cum = 1.0f;
            float res = 0.0f;
            for (int k = 0; k < K; ++k) {
                const int32_t n = pi[k];
                if (n < 0) continue;
                const float alpha = 1.0f - pd[k] / rr;
                res += (cum * alpha) * feats[(size_t)n * (size_t)C + (size_t)c];
                cum = cum * (1.0f - alpha);
            }
            if (pi[0] < 0) res = background[c];
            out[(size_t)pix * (size_t)C + (size_t)c] = res;
        }
    }
}
