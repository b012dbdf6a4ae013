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
 * Inflate for whole zlib streams of known decoded size (RFC 1950 / 1951), about twice as fast as zlib 1.2's on this
 * data: a 64-bit bit buffer refilled with one unaligned load, an 11-bit first-level table for the literal / length code
 * (8-bit for distances) with second-level tables for the longer codes, and matches copied eight bytes at a time.
 * It is an accelerator, not a replacement: anything it does not like (a malformed or unusual stream, a size or Adler-32
 * mismatch) makes it return non-zero, and the caller hands the same bytes to zlib, whose verdict stands.
 * `out` must have EPC_INFLATE_SLACK writable bytes beyond n_out (the match copy overshoots).
 * ------------------------------------------------------------------------------------------------------------------ */
#define EPC_INFLATE_SLACK 64
#define EPC_LL_ROOT 11
#define EPC_D_ROOT 8
#define EPC_F_LIT 0x1000u
#define EPC_F_LEN 0x2000u
#define EPC_F_EOB 0x4000u
#define EPC_F_SUB 0x8000u

static uint64_t rd_le64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

static uint32_t adler32_scalar(uint32_t adler, const uint8_t* p, size_t n) {
    uint32_t a = adler & 0xFFFF, b = adler >> 16;
    while (n > 0) {
        size_t k = n < 5552 ? n : 5552;  /* the largest run for which b cannot overflow 32 bits */
        n -= k;
        while (k--) { a += *p++; b += a; }
        a %= 65521u;
        b %= 65521u;
    }
    return (b << 16) | a;
}
#if defined(__x86_64__) && defined(__GNUC__)
#include <tmmintrin.h>
__attribute__((target("ssse3"))) static uint32_t adler32_ssse3(uint32_t adler, const uint8_t* p, size_t n) {
    uint32_t a = adler & 0xFFFF, b = adler >> 16;
    const __m128i w1 = _mm_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17);
    const __m128i w2 = _mm_setr_epi8(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m128i zero = _mm_setzero_si128(), ones = _mm_set1_epi16(1);
    while (n >= 32) {
        size_t blocks = n / 32;
        if (blocks > 5552 / 32) blocks = 5552 / 32;
        n -= blocks * 32;
        /* over one 32-byte block: a' = a + sum(x), b' = b + 32 a + sum((32 - i) x_i) */
        __m128i v_ps = _mm_set_epi32(0, 0, 0, (int)(a * (uint32_t)blocks));  /* sum of a before each block */
        __m128i v_a = _mm_setzero_si128(), v_b = _mm_set_epi32(0, 0, 0, (int)b);
        do {
            const __m128i x1 = _mm_loadu_si128((const __m128i*)p), x2 = _mm_loadu_si128((const __m128i*)(p + 16));
            v_ps = _mm_add_epi32(v_ps, v_a);
            v_a = _mm_add_epi32(v_a, _mm_add_epi32(_mm_sad_epu8(x1, zero), _mm_sad_epu8(x2, zero)));
            v_b = _mm_add_epi32(v_b, _mm_madd_epi16(_mm_maddubs_epi16(x1, w1), ones));
            v_b = _mm_add_epi32(v_b, _mm_madd_epi16(_mm_maddubs_epi16(x2, w2), ones));
            p += 32;
        } while (--blocks);
        v_b = _mm_add_epi32(v_b, _mm_slli_epi32(v_ps, 5));
        v_a = _mm_add_epi32(v_a, _mm_shuffle_epi32(v_a, 0x4E));
        v_a = _mm_add_epi32(v_a, _mm_shuffle_epi32(v_a, 0xB1));
        v_b = _mm_add_epi32(v_b, _mm_shuffle_epi32(v_b, 0x4E));
        v_b = _mm_add_epi32(v_b, _mm_shuffle_epi32(v_b, 0xB1));
        a = (a + (uint32_t)_mm_cvtsi128_si32(v_a)) % 65521u;
        b = (uint32_t)_mm_cvtsi128_si32(v_b) % 65521u;
    }
    return adler32_scalar((b << 16) | a, p, n);
}
static uint32_t adler32_fast(const uint8_t* p, size_t n) {
    return __builtin_cpu_supports("ssse3") ? adler32_ssse3(1u, p, n) : adler32_scalar(1u, p, n);
}
#else
static uint32_t adler32_fast(const uint8_t* p, size_t n) { return adler32_scalar(1u, p, n); }
#endif

static uint32_t bit_reverse(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

/* Decoding table of a canonical Huffman code.  lens[n_sym]: code lengths (0 = unused); root: bits of the first level;
 * base / extra: value and extra-bit count per symbol class as packed by the caller in sym_entry[] (flags | extra << 8 |
 * value << 16, without the length).  Returns 0, or -1 for an over-subscribed or incomplete code (one code of length 1 is
 * let through, as deflate allows for distances; its unused half stays invalid). */
static int build_table(const uint8_t* lens, int n_sym, const uint32_t* sym_entry, int root, uint32_t* table,
                       int table_cap) {
    int count[16] = {0}, maxlen = 0, used = 0;
    for (int i = 0; i < n_sym; ++i) { count[lens[i]]++; if (lens[i]) { used++; if (lens[i] > maxlen) maxlen = lens[i]; } }
    const int prim = 1 << root;
    for (int i = 0; i < prim; ++i) table[i] = 0;
    if (used == 0) return 0;  /* no code at all: every lookup hits an invalid entry */
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left = (left << 1) - count[l]; if (left < 0) return -1; }
    if (left > 0 && !(used == 1 && count[1] == 1)) return -1;
    uint32_t next_code[16], code = 0;
    count[0] = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
    const int sub_bits = maxlen > root ? maxlen - root : 0;
    int next_free = prim;
    for (int sym = 0; sym < n_sym; ++sym) {
        const int l = lens[sym];
        if (l == 0) continue;
        const uint32_t rev = bit_reverse(next_code[l]++, l);
        if (l <= root) {
            const uint32_t e = sym_entry[sym] | (uint32_t)l;
            for (uint32_t k = rev; k < (uint32_t)prim; k += 1u << l) table[k] = e;
        } else {
            const uint32_t pre = rev & (uint32_t)(prim - 1);
            if (table[pre] == 0) {
                if (next_free + (1 << sub_bits) > table_cap) return -1;
                table[pre] = EPC_F_SUB | ((uint32_t)sub_bits << 8) | (uint32_t)root | ((uint32_t)next_free << 16);
                for (int k = 0; k < (1 << sub_bits); ++k) table[next_free + k] = 0;
                next_free += 1 << sub_bits;
            }
            uint32_t* sub = table + (table[pre] >> 16);
            const uint32_t e = sym_entry[sym] | (uint32_t)(l - root);
            for (uint32_t k = rev >> root; k < (1u << sub_bits); k += 1u << (l - root)) sub[k] = e;
        }
    }
    return 0;
}

#define EPC_LL_CAP ((1 << EPC_LL_ROOT) + 288 * 16)
#define EPC_D_CAP ((1 << EPC_D_ROOT) + 32 * 128)

typedef struct {
    uint32_t ll[EPC_LL_CAP];
    uint32_t d[EPC_D_CAP];
} inflate_tables;

static void symbol_entries(uint32_t* ll_e, uint32_t* d_e) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83,
                                       99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025,
                                       1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12,
                                     12, 13, 13};
    for (int i = 0; i < 256; ++i) ll_e[i] = EPC_F_LIT | ((uint32_t)i << 16);
    ll_e[256] = EPC_F_EOB;
    for (int i = 0; i < 29; ++i) ll_e[257 + i] = EPC_F_LEN | ((uint32_t)lext[i] << 8) | ((uint32_t)lbase[i] << 16);
    ll_e[286] = ll_e[287] = 0;  /* never valid in a stream; their codes exist in the fixed code only */
    for (int i = 0; i < 30; ++i) d_e[i] = EPC_F_LEN | ((uint32_t)dext[i] << 8) | ((uint32_t)dbase[i] << 16);
    d_e[30] = d_e[31] = 0;
}

/* 0 on success: exactly n_out bytes written, the stream ended where the input ends, Adler-32 matches. */
static int fast_inflate_zlib(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out) {
    if (n_in < 6) return -1;
    if ((in[0] & 0x0F) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return -1;
    const uint8_t* const in_end = in + n_in - 4;  /* the Adler-32 trailer is not part of the deflate data */
    const uint8_t* ip = in + 2;
    uint8_t* op = out;
    uint8_t* const out_end = out + n_out;
    uint64_t bb = 0;
    int bc = 0;
    uint32_t ll_e[288], d_e[32];
    symbol_entries(ll_e, d_e);
    inflate_tables* T = (inflate_tables*)malloc(sizeof(inflate_tables));
    if (!T) return -1;
    int rc = -1, last = 0;
/* bytes past the end read as zero; a stream that really needs them is caught by the final position check */
#define EPC_REFILL()                                                                      \
    do {                                                                                  \
        if (in_end - ip >= 8) {                                                           \
            bb |= rd_le64(ip) << bc;                                                      \
            ip += (63 - bc) >> 3;                                                         \
            bc |= 56;                                                                     \
        } else {                                                                          \
            while (bc <= 56) {                                                            \
                if (ip < in_end) bb |= (uint64_t)(*ip) << bc;                             \
                ++ip;                                                                     \
                bc += 8;                                                                  \
            }                                                                             \
        }                                                                                 \
    } while (0)
#define EPC_TAKE(n_) (bb >>= (n_), bc -= (n_))
    while (!last) {
        EPC_REFILL();
        last = (int)(bb & 1u);
        const int type = (int)((bb >> 1) & 3u);
        EPC_TAKE(3);
        if (type == 0) {  /* stored: back to a byte boundary, LEN, ~LEN, bytes */
            EPC_TAKE(bc & 7);
            /* give whole unread bytes back to the input */
            ip -= bc >> 3;
            bb = 0;
            bc = 0;
            if (ip > in_end || in_end - ip < 4) goto done;
            const uint32_t len = ip[0] | ((uint32_t)ip[1] << 8), nlen = ip[2] | ((uint32_t)ip[3] << 8);
            ip += 4;
            if ((len ^ nlen) != 0xFFFFu || (size_t)(in_end - ip) < len || (size_t)(out_end - op) < len) goto done;
            memcpy(op, ip, len);
            op += len;
            ip += len;
            continue;
        }
        if (type == 3) goto done;
        uint8_t lens[288 + 32];
        int n_ll, n_d;
        if (type == 1) {
            n_ll = 288;
            n_d = 32;
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        } else {
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            n_ll = (int)(bb & 31u) + 257;
            n_d = (int)((bb >> 5) & 31u) + 1;
            const int n_cl = (int)((bb >> 10) & 15u) + 4;
            EPC_TAKE(14);
            if (n_ll > 286 || n_d > 30) goto done;
            uint8_t cl[19] = {0};
            for (int i = 0; i < n_cl; ++i) {
                if (bc < 3) EPC_REFILL();
                cl[order[i]] = (uint8_t)(bb & 7u);
                EPC_TAKE(3);
            }
            uint32_t cl_e[19], cl_t[128];
            for (int i = 0; i < 19; ++i) cl_e[i] = EPC_F_LIT | ((uint32_t)i << 16);
            if (build_table(cl, 19, cl_e, 7, cl_t, 128) != 0) goto done;
            int i = 0;
            while (i < n_ll + n_d) {
                EPC_REFILL();
                const uint32_t e = cl_t[bb & 127u];
                if (!(e & EPC_F_LIT)) goto done;
                EPC_TAKE((int)(e & 0xFFu));
                const int sym = (int)(e >> 16);
                if (sym < 16) {
                    lens[i++] = (uint8_t)sym;
                } else {
                    int rep, val = 0;
                    if (sym == 16) {
                        if (i == 0) goto done;
                        val = lens[i - 1];
                        rep = 3 + (int)(bb & 3u);
                        EPC_TAKE(2);
                    } else if (sym == 17) {
                        rep = 3 + (int)(bb & 7u);
                        EPC_TAKE(3);
                    } else {
                        rep = 11 + (int)(bb & 127u);
                        EPC_TAKE(7);
                    }
                    if (i + rep > n_ll + n_d) goto done;
                    while (rep--) lens[i++] = (uint8_t)val;
                }
            }
            if (lens[256] == 0) goto done;  /* no end-of-block code */
            /* the distance lengths follow the literal / length ones: move them to their fixed place */
            memmove(lens + 288, lens + n_ll, (size_t)n_d);
            for (int k = n_ll; k < 288; ++k) lens[k] = 0;
        }
        if (build_table(lens, n_ll, ll_e, EPC_LL_ROOT, T->ll, EPC_LL_CAP) != 0) goto done;
        if (build_table(lens + 288, n_d, d_e, EPC_D_ROOT, T->d, EPC_D_CAP) != 0) goto done;
        for (;;) {
            EPC_REFILL();
            uint32_t e = T->ll[bb & ((1u << EPC_LL_ROOT) - 1u)];
            if (e & EPC_F_SUB) {
                EPC_TAKE(EPC_LL_ROOT);
                e = T->ll[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))];
            }
            EPC_TAKE((int)(e & 0xFFu));
            if (e & EPC_F_LIT) {
                if (op >= out_end) goto done;
                *op++ = (uint8_t)(e >> 16);
                /* a second literal from the same refill, if the next code is one (the bit budget allows 15 + 15 more) */
                uint32_t e2 = T->ll[bb & ((1u << EPC_LL_ROOT) - 1u)];
                if ((e2 & EPC_F_LIT) && op < out_end) {
                    EPC_TAKE((int)(e2 & 0xFFu));
                    *op++ = (uint8_t)(e2 >> 16);
                }
                continue;
            }
            if (e & EPC_F_EOB) break;
            if (!(e & EPC_F_LEN)) goto done;
            const int xb = (int)((e >> 8) & 15u);
            const size_t len = (size_t)(e >> 16) + (size_t)(bb & ((1u << xb) - 1u));
            EPC_TAKE(xb);
            if (bc < 32) EPC_REFILL();
            uint32_t de = T->d[bb & ((1u << EPC_D_ROOT) - 1u)];
            if (de & EPC_F_SUB) {
                EPC_TAKE(EPC_D_ROOT);
                de = T->d[(de >> 16) + (uint32_t)(bb & ((1u << ((de >> 8) & 15u)) - 1u))];
            }
            if (!(de & EPC_F_LEN)) goto done;
            EPC_TAKE((int)(de & 0xFFu));
            const int dxb = (int)((de >> 8) & 15u);
            const size_t dist = (size_t)(de >> 16) + (size_t)(bb & ((1u << dxb) - 1u));
            EPC_TAKE(dxb);
            if (dist > (size_t)(op - out) || len > (size_t)(out_end - op)) goto done;
            const uint8_t* src = op - dist;
            if (dist >= 8) {  /* eight bytes at a time; overshoots by up to 7 bytes into the slack */
                uint8_t* dst = op;
                uint8_t* const stop = op + len;
                do {
                    memcpy(dst, src, 8);
                    dst += 8;
                    src += 8;
                } while (dst < stop);
            } else if (dist == 1) {
                memset(op, *src, len);
            } else {
                for (size_t k = 0; k < len; ++k) op[k] = src[k];
            }
            op += len;
        }
    }
    /* whole bytes still in the bit buffer were not consumed: the deflate data must end exactly at the trailer */
    ip -= bc >> 3;
    if (ip != in_end || op != out_end) goto done;
    {
        const uint32_t want = ((uint32_t)in_end[0] << 24) | ((uint32_t)in_end[1] << 16) | ((uint32_t)in_end[2] << 8) | in_end[3];
        if (adler32_fast(out, n_out) != want) goto done;
    }
    rc = 0;
done:
    free(T);
    return rc;
#undef EPC_REFILL
#undef EPC_TAKE
}

/* zlib stream -> exactly n_out bytes; the fast decoder first, zlib itself when that declines. */
static int inflate_exact(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out) {
    if (fast_inflate_zlib(in, n_in, out, n_out) == 0) return EPC_OK;
    if (n_out == 0) {  /* uncompress() treats a zero-length destination as "discard one byte": give it a real byte */
        uint8_t one;
        uLongf got0 = 1;
        return (uncompress(&one, &got0, in, (uLong)n_in) == Z_OK && got0 == 0) ? EPC_OK : EPC_E_CORRUPT;
    }
    uLongf got = (uLongf)n_out;
    if (uncompress(out, &got, in, (uLong)n_in) != Z_OK || (size_t)got != n_out) return EPC_E_CORRUPT;
    return EPC_OK;
}

int epc_inflate_exact(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out, int32_t fast_only) {
    /* test hook: out needs EPC_INFLATE_SLACK spare bytes; fast_only != 0 reports whether the fast decoder accepted */
    if (fast_only) return fast_inflate_zlib(in, n_in, out, n_out);
    return inflate_exact(in, n_in, out, n_out);
}

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
 * split (first half of the buffer: bytes of even position) that OpenEXR applies before deflate / RLE.
 * The predictor is a running sum modulo 256 of (t[i] - 128); with SSE2 sixteen bytes are summed in four shift-and-add
 * steps and the carry of the previous sixteen is added on top, and the two halves are interleaved with byte unpacks. */
#if defined(__SSE2__)
#include <emmintrin.h>
static void exr_unfilter(uint8_t* t, uint8_t* out, size_t n) {
    size_t i = 1;
    uint8_t carry = n ? t[0] : 0;
    const __m128i bias = _mm_set1_epi8((char)128);
    for (; i + 16 <= n; i += 16) {
        __m128i x = _mm_sub_epi8(_mm_loadu_si128((const __m128i*)(t + i)), bias);
        x = _mm_add_epi8(x, _mm_slli_si128(x, 1));
        x = _mm_add_epi8(x, _mm_slli_si128(x, 2));
        x = _mm_add_epi8(x, _mm_slli_si128(x, 4));
        x = _mm_add_epi8(x, _mm_slli_si128(x, 8));
        x = _mm_add_epi8(x, _mm_set1_epi8((char)carry));
        _mm_storeu_si128((__m128i*)(t + i), x);
        carry = t[i + 15];
    }
    for (; i < n; ++i) t[i] = (uint8_t)(t[i - 1] + t[i] - 128);
    const size_t half = (n + 1) / 2, pairs = n / 2;
    const uint8_t *a = t, *b = t + half;
    size_t k = 0;
    for (; k + 16 <= pairs; k += 16) {
        const __m128i va = _mm_loadu_si128((const __m128i*)(a + k)), vb = _mm_loadu_si128((const __m128i*)(b + k));
        _mm_storeu_si128((__m128i*)(out + 2 * k), _mm_unpacklo_epi8(va, vb));
        _mm_storeu_si128((__m128i*)(out + 2 * k + 16), _mm_unpackhi_epi8(va, vb));
    }
    for (; k < pairs; ++k) {
        out[2 * k] = a[k];
        out[2 * k + 1] = b[k];
    }
    if (n & 1) out[n - 1] = a[half - 1];
}
#else
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
#endif

int epc_exr_half_z_decode(const uint8_t* buf, size_t n, uint16_t* out, int32_t H, int32_t W) {
    exr_info info;
    int rc = exr_parse(buf, n, &info);
    if (rc != EPC_OK) return rc;
    if (W != info.xmax - info.xmin + 1 || H != info.ymax - info.ymin + 1) return EPC_E_SIZE;
    const int lines = info.lines_per_chunk;
    const size_t n_chunks = ((size_t)H + (size_t)lines - 1) / (size_t)lines;
    if (info.table + 8 * n_chunks > n) return EPC_E_CORRUPT;
    const size_t max_raw = (size_t)lines * info.line_bytes;
    uint8_t* tmp = (uint8_t*)malloc(2 * max_raw + 2 * EPC_INFLATE_SLACK);
    if (!tmp) return EPC_E_NOMEM;
    uint8_t* tmp2 = tmp + max_raw + EPC_INFLATE_SLACK;
    for (size_t c = 0; c < n_chunks && rc == EPC_OK; ++c) {
        /* the offset and the chunk size come from the file: every comparison is written so that it cannot wrap */
        const uint64_t off = rd_u64(buf + info.table + 8 * c);
        if (n < 8 || off > (uint64_t)n - 8) { rc = EPC_E_CORRUPT; break; }
        const int32_t y = rd_i32(buf + off), size = rd_i32(buf + off + 4);
        /* the table lists the chunks by increasing row whatever the file's line order: chunk c starts at row ymin + c * lines
         * (so every output row is written exactly once) */
        if ((int64_t)y != (int64_t)info.ymin + (int64_t)c * lines || size < 0 || (uint64_t)size > (uint64_t)n - off - 8) {
            rc = EPC_E_CORRUPT;
            break;
        }
        int rows = info.ymax - y + 1;
        if (rows > lines) rows = lines;
        const size_t raw_size = (size_t)rows * info.line_bytes;
        if ((size_t)size > raw_size) { rc = EPC_E_CORRUPT; break; }
        const uint8_t* data = buf + off + 8;
        const uint8_t* raw;
        if ((size_t)size == raw_size) { /* uncompressed file, or a chunk that did not shrink and is stored as it is */
            raw = data;
        } else if (info.compression == 0) { /* an uncompressed chunk holds exactly its rows: the row copy reads them all */
            rc = EPC_E_CORRUPT;
            break;
        } else {
            if (info.compression == 1) {
                rc = exr_unrle(data, (size_t)size, tmp, raw_size);
            } else {
                rc = inflate_exact(data, (size_t)size, tmp, raw_size);
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
    uint8_t* raw = (uint8_t*)malloc(raw_size + EPC_INFLATE_SLACK);
    if (!raw) return EPC_E_NOMEM;
    /* the IDAT chunks hold one zlib stream, cut into pieces: join them (a single chunk is used where it lies) */
    size_t pos = 8, idat_total = 0, idat_chunks = 0;
    const uint8_t* first = NULL;
    while (pos + 12 <= n) {
        const uint32_t len = rd_be32(buf + pos);
        if (pos + 12 + (size_t)len > n) { rc = EPC_E_CORRUPT; break; }
        if (memcmp(buf + pos + 4, "IDAT", 4) == 0) {
            if (idat_chunks++ == 0) first = buf + pos + 8;
            idat_total += len;
        } else if (memcmp(buf + pos + 4, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (rc == EPC_OK && idat_chunks == 0) rc = EPC_E_CORRUPT;
    if (rc == EPC_OK) {
        uint8_t* joined = NULL;
        const uint8_t* stream = first;
        if (idat_chunks > 1) {
            joined = (uint8_t*)malloc(idat_total ? idat_total : 1);
            if (!joined) { free(raw); return EPC_E_NOMEM; }
            size_t o = 0;
            for (pos = 8; pos + 12 <= n;) {
                const uint32_t len = rd_be32(buf + pos);
                if (memcmp(buf + pos + 4, "IDAT", 4) == 0) { memcpy(joined + o, buf + pos + 8, len); o += len; }
                else if (memcmp(buf + pos + 4, "IEND", 4) == 0) break;
                pos += 12 + (size_t)len;
            }
            stream = joined;
        }
        rc = inflate_exact(stream, idat_total, raw, raw_size);
        free(joined);
    }
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
