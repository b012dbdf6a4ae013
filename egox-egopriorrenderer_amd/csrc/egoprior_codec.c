/*
 * Host-side decoders of the two per-frame containers of the ViPE artifacts, for the ingestion side of the renderer:
 *   depth : OpenEXR, single-part scan-line file with one HALF channel named Z, compression none / RLE / ZIPS / ZIP
 *           (what vipe/utils/io.py:270-279 writes and :282-312 reads through the OpenEXR library)
 *   masks : PNG, 8-bit greyscale, non-interlaced (what vipe/utils/io.py:353-370 writes with cv2.imwrite)
 * Both decode straight into a caller-owned buffer (page-locked staging memory of the upload ring, for instance), so a
 * frame costs one inflate and one pass over its bytes.  Anything outside these subsets returns a negative code and the
 * Python decoders of egoprior/io.py (same formats, slower) take over.  Thread-safe: no global state.
 *
 * Build: gcc -O3 -fPIC -shared egoprior_codec.c -lz   (see Makefile)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define EPC_OK 0
#define EPC_E_FORMAT -1     /* not a file of the supported subset */
#define EPC_E_CORRUPT -2    /* truncated or inconsistent data */
#define EPC_E_SIZE -3       /* output buffer geometry does not match the file */
#define EPC_E_NOMEM -4

static int32_t rd_i32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
static uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd_u64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd_be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* ------------------------------------------------------------------------------------------------------------------
 * OpenEXR
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t xmin, ymin, xmax, ymax;
    int compression;      /* 0 none, 1 RLE, 2 ZIPS, 3 ZIP */
    int lines_per_chunk;
    size_t line_bytes;    /* bytes of one scan line over all channels */
    size_t z_off;         /* byte offset of channel Z inside a scan line */
    size_t table;         /* file offset of the chunk offset table */
} exr_info;

static int exr_parse(const uint8_t* buf, size_t n, exr_info* info) {
    if (n < 16) return EPC_E_FORMAT;
    const int32_t magic = rd_i32(buf);
    const uint32_t version = rd_u32(buf + 4);
    if (magic != 20000630 || (version & 0xFF) != 2 || (version & 0x1A00)) return EPC_E_FORMAT; /* tiled / deep / multi-part */
    size_t pos = 8;
    int have_window = 0, have_comp = 0, have_z = 0;
    size_t bytes_per_pixel = 0, z_pix_off = 0;
    while (pos < n && buf[pos] != 0) {
        const uint8_t* name = buf + pos;
        const uint8_t* e = memchr(name, 0, n - pos);
        if (!e) return EPC_E_CORRUPT;
        const uint8_t* type = e + 1;
        const uint8_t* e2 = memchr(type, 0, n - (size_t)(type - buf));
        if (!e2 || (size_t)(e2 - buf) + 5 > n) return EPC_E_CORRUPT;
        const int32_t size = rd_i32(e2 + 1);
        const uint8_t* val = e2 + 5;
        if (size < 0 || (size_t)(val - buf) + (size_t)size > n) return EPC_E_CORRUPT;
        if (strcmp((const char*)name, "dataWindow") == 0 && size == 16) {
            info->xmin = rd_i32(val); info->ymin = rd_i32(val + 4); info->xmax = rd_i32(val + 8); info->ymax = rd_i32(val + 12);
            have_window = 1;
        } else if (strcmp((const char*)name, "compression") == 0 && size >= 1) {
            info->compression = val[0];
            have_comp = 1;
        } else if (strcmp((const char*)name, "channels") == 0) {
            size_t q = 0;
            while (q < (size_t)size && val[q] != 0) {
                const uint8_t* ce = memchr(val + q, 0, (size_t)size - q);
                if (!ce || (size_t)(ce - val) + 17 > (size_t)size) return EPC_E_CORRUPT;
                const int32_t ptype = rd_i32(ce + 1);
                const int32_t xs = rd_i32(ce + 9), ys = rd_i32(ce + 13);
                if (xs != 1 || ys != 1) return EPC_E_FORMAT; /* subsampled channels */
                const size_t bpp = (ptype == 1) ? 2 : 4;
                if (strcmp((const char*)(val + q), "Z") == 0) {
                    if (ptype != 1) return EPC_E_FORMAT; /* Z must be HALF */
                    have_z = 1;
                    z_pix_off = bytes_per_pixel;
                }
                bytes_per_pixel += bpp;
                q = (size_t)(ce - val) + 17;
            }
        }
        pos = (size_t)(val - buf) + (size_t)size;
    }
    if (pos >= n || !have_window || !have_comp || !have_z) return EPC_E_FORMAT;
    pos += 1;
    if (info->compression < 0 || info->compression > 3) return EPC_E_FORMAT;
    info->lines_per_chunk = (info->compression == 3) ? 16 : 1;
    const int64_t W = (int64_t)info->xmax - info->xmin + 1, H = (int64_t)info->ymax - info->ymin + 1;
    if (W <= 0 || H <= 0 || W > 65536 || H > 65536) return EPC_E_FORMAT;
    info->line_bytes = bytes_per_pixel * (size_t)W;
    info->z_off = z_pix_off * (size_t)W; /* channels are stored one after the other within a scan line */
    info->table = pos;
    return EPC_OK;
}

int epc_exr_half_z_size(const uint8_t* buf, size_t n, int32_t* H, int32_t* W) {
    exr_info info;
    const int rc = exr_parse(buf, n, &info);
    if (rc != EPC_OK) return rc;
    *W = info.xmax - info.xmin + 1;
    *H = info.ymax - info.ymin + 1;
    return EPC_OK;
}

/* OpenEXR's run-length code: a count byte c >= 0 repeats the next byte c + 1 times, c < 0 copies -c bytes. */
static int exr_unrle(const uint8_t* src, size_t n_in, uint8_t* dst, size_t n_out) {
    size_t i = 0, o = 0;
    while (i < n_in && o < n_out) {
        const int c = (int8_t)src[i++];
        if (c < 0) {
            const size_t k = (size_t)(-c);
            if (i + k > n_in || o + k > n_out) return EPC_E_CORRUPT;
            memcpy(dst + o, src + i, k);
            i += k;
            o += k;
        } else {
            const size_t k = (size_t)c + 1;
            if (i >= n_in || o + k > n_out) return EPC_E_CORRUPT;
            memset(dst + o, src[i++], k);
            o += k;
        }
    }
    return o == n_out ? EPC_OK : EPC_E_CORRUPT;
}

/* Undo the byte predictor (each byte is stored as its difference to the previous one plus 128) and the even / odd byte
 * split (first half of the buffer: bytes of even position) that OpenEXR applies before deflate / RLE. */
static void exr_unfilter(uint8_t* t, uint8_t* out, size_t n) {
    for (size_t i = 1; i < n; ++i) t[i] = (uint8_t)(t[i - 1] + t[i] - 128);
    const size_t half = (n + 1) / 2;
    const uint8_t *a = t, *b = t + half;
    size_t o = 0;
    for (size_t i = 0; i < n / 2; ++i) {
        out[o++] = a[i];
        out[o++] = b[i];
    }
    if (n & 1) out[o] = a[half - 1];
}

int epc_exr_half_z_decode(const uint8_t* buf, size_t n, uint16_t* out, int32_t H, int32_t W) {
    exr_info info;
    int rc = exr_parse(buf, n, &info);
    if (rc != EPC_OK) return rc;
    if (W != info.xmax - info.xmin + 1 || H != info.ymax - info.ymin + 1) return EPC_E_SIZE;
    const int lines = info.lines_per_chunk;
    const size_t n_chunks = ((size_t)H + (size_t)lines - 1) / (size_t)lines;
    if (info.table + 8 * n_chunks > n) return EPC_E_CORRUPT;
    const size_t max_raw = (size_t)lines * info.line_bytes;
    uint8_t* tmp = (uint8_t*)malloc(2 * max_raw + 16);
    if (!tmp) return EPC_E_NOMEM;
    uint8_t* tmp2 = tmp + max_raw + 8;
    for (size_t c = 0; c < n_chunks && rc == EPC_OK; ++c) {
        const uint64_t off = rd_u64(buf + info.table + 8 * c);
        if (off + 8 > n) { rc = EPC_E_CORRUPT; break; }
        const int32_t y = rd_i32(buf + off), size = rd_i32(buf + off + 4);
        if (y < info.ymin || y > info.ymax || size < 0 || off + 8 + (uint64_t)size > n) { rc = EPC_E_CORRUPT; break; }
        int rows = info.ymax - y + 1;
        if (rows > lines) rows = lines;
        const size_t raw_size = (size_t)rows * info.line_bytes;
        if ((size_t)size > raw_size) { rc = EPC_E_CORRUPT; break; }
        const uint8_t* data = buf + off + 8;
        const uint8_t* raw;
        if (info.compression == 0 || (size_t)size == raw_size) { /* a chunk that does not shrink is stored as it is */
            raw = data;
        } else {
            if (info.compression == 1) {
                rc = exr_unrle(data, (size_t)size, tmp, raw_size);
            } else {
                uLongf got = (uLongf)raw_size;
                if (uncompress(tmp, &got, data, (uLong)size) != Z_OK || (size_t)got != raw_size) rc = EPC_E_CORRUPT;
            }
            if (rc != EPC_OK) break;
            exr_unfilter(tmp, tmp2, raw_size);
            raw = tmp2;
        }
        for (int r = 0; r < rows; ++r)
            memcpy(out + (size_t)(y - info.ymin + r) * (size_t)W, raw + (size_t)r * info.line_bytes + info.z_off, 2 * (size_t)W);
    }
    free(tmp);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * PNG, 8-bit greyscale
 * ------------------------------------------------------------------------------------------------------------------ */
static int png_header(const uint8_t* buf, size_t n, int32_t* H, int32_t* W) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (n < 33 || memcmp(buf, sig, 8) != 0) return EPC_E_FORMAT;
    if (rd_be32(buf + 8) != 13 || memcmp(buf + 12, "IHDR", 4) != 0) return EPC_E_FORMAT;
    const uint32_t w = rd_be32(buf + 16), h = rd_be32(buf + 20);
    const int depth = buf[24], ctype = buf[25], interlace = buf[28];
    if (depth != 8 || ctype != 0 || interlace != 0) return EPC_E_FORMAT; /* the Python decoder (or OpenCV) takes the rest */
    if (w == 0 || h == 0 || w > 65536 || h > 65536) return EPC_E_FORMAT;
    *W = (int32_t)w;
    *H = (int32_t)h;
    return EPC_OK;
}

int epc_png_gray8_size(const uint8_t* buf, size_t n, int32_t* H, int32_t* W) { return png_header(buf, n, H, W); }

static inline int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

int epc_png_gray8_decode(const uint8_t* buf, size_t n, uint8_t* out, int32_t H, int32_t W) {
    int32_t h = 0, w = 0;
    int rc = png_header(buf, n, &h, &w);
    if (rc != EPC_OK) return rc;
    if (h != H || w != W) return EPC_E_SIZE;
    const size_t stride = (size_t)W, raw_size = ((size_t)W + 1) * (size_t)H;
    uint8_t* raw = (uint8_t*)malloc(raw_size);
    if (!raw) return EPC_E_NOMEM;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) { free(raw); return EPC_E_NOMEM; }
    zs.next_out = raw;
    zs.avail_out = (uInt)raw_size;
    size_t pos = 8;
    int done = 0;
    while (pos + 12 <= n && !done) { /* the IDAT chunks hold one zlib stream, cut into pieces */
        const uint32_t len = rd_be32(buf + pos);
        if (pos + 12 + (size_t)len > n) { rc = EPC_E_CORRUPT; break; }
        if (memcmp(buf + pos + 4, "IDAT", 4) == 0) {
            zs.next_in = (Bytef*)(buf + pos + 8);
            zs.avail_in = len;
            const int zr = inflate(&zs, Z_NO_FLUSH);
            if (zr == Z_STREAM_END) done = 1;
            else if (zr != Z_OK && zr != Z_BUF_ERROR) { rc = EPC_E_CORRUPT; break; }
        } else if (memcmp(buf + pos + 4, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    const size_t got = raw_size - zs.avail_out;
    inflateEnd(&zs);
    if (rc == EPC_OK && got != raw_size) rc = EPC_E_CORRUPT;
    for (int32_t y = 0; y < H && rc == EPC_OK; ++y) {
        const uint8_t* line = raw + (size_t)y * (stride + 1);
        const int ft = line[0];
        const uint8_t* in = line + 1;
        uint8_t* cur = out + (size_t)y * stride;
        const uint8_t* prev = (y > 0) ? cur - stride : NULL;
        switch (ft) {
            case 0:
                memcpy(cur, in, stride);
                break;
            case 1: { /* Sub */
                uint8_t left = 0;
                for (size_t x = 0; x < stride; ++x) { left = (uint8_t)(in[x] + left); cur[x] = left; }
                break;
            }
            case 2: /* Up */
                for (size_t x = 0; x < stride; ++x) cur[x] = (uint8_t)(in[x] + (prev ? prev[x] : 0));
                break;
            case 3: { /* Average */
                int left = 0;
                for (size_t x = 0; x < stride; ++x) {
                    const int up = prev ? prev[x] : 0;
                    left = (in[x] + ((left + up) >> 1)) & 255;
                    cur[x] = (uint8_t)left;
                }
                break;
            }
            case 4: { /* Paeth */
                int left = 0, ul = 0;
                for (size_t x = 0; x < stride; ++x) {
                    const int up = prev ? prev[x] : 0;
                    left = (in[x] + paeth(left, up, ul)) & 255;
                    cur[x] = (uint8_t)left;
                    ul = up;
                }
                break;
            }
            default:
                rc = EPC_E_CORRUPT;
        }
    }
    free(raw);
    return rc;
}
