/* dither_host.c -- a stand-in for the part of the C host that owns the dither tables.
 *
 * convolver_cbuf2raw(apply_dither = 1) reads three globals of the host program, exactly like the
 * reference's real2raw path does (dither.h:24-26, dither_funs.h): `dither_randtab`,
 * `dither_randtab_size`, `dither_randmap`, plus the caller's `struct dither_state`
 * (dither.h:17-22).  libbfhip.so refers to them as weak symbols, so they have to exist in the
 * PROGRAM that links it -- which is why this test is a small C program and not a ctypes call.
 *
 * The table bytes come from the golden file (they were produced by the reference's dither_init);
 * randmap is the linear map dither_init documents: index d in [-256, 254], value
 * 0.5 + (d + 1) / 255 with the two end points pinned to -0.5 and 1.5.
 *
 * usage: dither_host <in.bin> <out.bin>
 *   in : int32 rs, L, n_ch, n_blk, table_size, spacing; int8 table[table_size];
 *        real x[n_blk][n_ch][L]
 *   out: int16 raw[n_blk][n_ch][L]; int32 randtab_ptr[n_blk][n_ch]; double of[n_ch][4]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bfhip_convolver.h"

int8_t *dither_randtab;
int dither_randtab_size;
void *dither_randmap;

static void die(const char *m) { fprintf(stderr, "dither_host: %s\n", m); exit(2); }

int main(int argc, char **argv)
{
    int32_t hdr[6];
    FILE *f;
    int rs, L, n_ch, n_blk, spacing, b, c, d;
    size_t n_x;
    void *x, *cbuf;
    int16_t *raw;
    int32_t *ptrs;
    double *ofs;
    struct bfhip_dither_state *ds;
    struct bfhip_overflow *of;
    struct bfhip_buffer_format bf;

    if (argc != 3) die("usage: dither_host in.bin out.bin");
    if ((f = fopen(argv[1], "rb")) == NULL) die("cannot open input");
    if (fread(hdr, sizeof(int32_t), 6, f) != 6) die("short header");
    rs = hdr[0]; L = hdr[1]; n_ch = hdr[2]; n_blk = hdr[3]; dither_randtab_size = hdr[4]; spacing = hdr[5];
    dither_randtab = malloc(dither_randtab_size);
    if (fread(dither_randtab, 1, dither_randtab_size, f) != (size_t)dither_randtab_size) die("short table");
    n_x = (size_t)n_blk * n_ch * L;
    x = malloc(n_x * rs);
    if (fread(x, rs, n_x, f) != n_x) die("short samples");
    fclose(f);

    dither_randmap = (uint8_t *)malloc((size_t)511 * rs) + (size_t)256 * rs;
    for (d = -256; d <= 254; d++) {
        if (rs == 4) {
            float v = d == -256 ? -0.5f : (d == 254 ? 1.5f : 0.5 + 1.0 / 255.0 + 1.0 / 255.0 * (float)d);
            ((float *)dither_randmap)[d] = v;
        } else {
            double v = d == -256 ? -0.5 : (d == 254 ? 1.5 : 0.5 + 1.0 / 255.0 + 1.0 / 255.0 * (double)d);
            ((double *)dither_randmap)[d] = v;
        }
    }

    if (!convolver_init(NULL, L, rs)) die("convolver_init failed");
    ds = calloc(n_ch, sizeof(*ds));
    of = calloc(n_ch, sizeof(*of));
    for (c = 0; c < n_ch; c++) {
        ds[c].randtab_ptr = c * spacing + 1;              /* dither_init's start offsets */
        of[c].max = 32767.0;
    }
    memset(&bf, 0, sizeof(bf));
    bf.sf.isfloat = 0; bf.sf.swap = 0; bf.sf.bytes = 2; bf.sf.sbytes = 2; bf.sf.scale = 1.0 / 32768.0;
    bf.sample_spacing = 1; bf.byte_offset = 0;
    raw = calloc(n_x, sizeof(int16_t));
    ptrs = calloc((size_t)n_blk * n_ch, sizeof(int32_t));
    cbuf = calloc((size_t)2 * L, rs);
    for (b = 0; b < n_blk; b++) {
        for (c = 0; c < n_ch; c++) {
            const size_t at = ((size_t)b * n_ch + c) * L;
            memcpy(cbuf, (uint8_t *)x + at * rs, (size_t)L * rs);
            convolver_cbuf2raw(cbuf, raw + at, &bf, 1, &ds[c], &of[c]);
            if (bfhip_convolver_last_fatal() != 0) die("convolver_cbuf2raw reported a fatal error");
            ptrs[b * n_ch + c] = ds[c].randtab_ptr;
        }
    }
    ofs = calloc((size_t)n_ch * 4, sizeof(double));
    for (c = 0; c < n_ch; c++) {
        ofs[4 * c + 0] = of[c].n_overflows; ofs[4 * c + 1] = of[c].intlargest;
        ofs[4 * c + 2] = of[c].largest; ofs[4 * c + 3] = of[c].max;
    }
    if ((f = fopen(argv[2], "wb")) == NULL) die("cannot open output");
    fwrite(raw, sizeof(int16_t), n_x, f);
    fwrite(ptrs, sizeof(int32_t), (size_t)n_blk * n_ch, f);
    fwrite(ofs, sizeof(double), (size_t)n_ch * 4, f);
    fclose(f);
    return 0;
}
