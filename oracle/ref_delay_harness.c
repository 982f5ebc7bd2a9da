/*
 * TEST INFRASTRUCTURE ONLY -- never linked into or called from the product.
 *
 * Second harness, second library (oracle/_ref/libbfref_delay.so): the reference's integer
 * sample delay (delay.c:78-340) and its Kaiser window (firwindow.c), compiled from the sources
 * where they lie under /root/reference and exported under "ref_*" names for tests/.
 *
 * delay.c also holds the sub-sample delay code, which calls convolver_td_new /
 * convolver_td_convolve / convolver_td_block_length (fftw_convolver.c -> FFTW3, absent here).
 * The entry points used below never reach them; oracle/Makefile binds the three names to
 * address 0 when it links THIS library (a call would fault at once).  The library that pins the
 * hot loop (libbfref.so, ref_harness.c) is linked without any such placeholder.
 */
#include <stdbool.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <inttypes.h>

#include "defs.h"
#include "bfconf.h"
#include "delay.h"
#include "firwindow.h"

static struct bfconf ref_bfconf_storage;
struct bfconf *bfconf = &ref_bfconf_storage;

void
bf_exit(int status)
{
    /* emalloc.c ends the program through this on an allocation failure */
    fprintf(stderr, "ref_delay_harness: bf_exit(%d)\n", status);
    abort();
}

void *
ref_delay_allocate(int fragment_size, int initdelay, int maxdelay, int sample_size)
{
    return delay_allocate_buffer(fragment_size, initdelay, maxdelay, sample_size);
}

void
ref_delay_update(void *db, void *buf, int sample_size, int sample_spacing, int delay, void *target)
{
    delay_update((delaybuffer_t *)db, buf, sample_size, sample_spacing, delay, target);
}

void
ref_firwindow_kaiser(void *target, int len, double offset, double beta, int realsize)
{
    firwindow_kaiser(target, len, offset, beta, realsize);
}
