/*
 * bfhip_nupc.h -- non-uniform partitioned convolution (low-latency first block) on top of the
 * uniform engines of bfhip.h.  An EXTENSION beyond the reference, which only supports uniform
 * partitions (`filter_length: L,N`, bfconf.c:1495-1520): BASELINE.json's room-correction
 * configuration asks for it.  Results are the linear convolution a uniform run computes; only
 * the I/O block -- the latency -- shrinks from L to the smallest segment length.
 *
 * The impulse response is cut into segments; segment k has partition length seg_length[k]
 * (ascending powers of two, each a multiple of the previous) and seg_blocks[k] partitions and
 * covers the taps after the previous segments.  I/O happens in blocks of seg_length[0] frames.
 * Segment k must start at a tap >= seg_length[k] - seg_length[0] so that its result is ready
 * when needed (checked); "2 x 64, 2 x 128, 2 x 256, ..." style schedules satisfy it.
 *
 * Scheduling: a segment whose first output frame is due later than the period it is launched in
 * (every segment but the first in the doubling schedule) runs on its own low-priority stream
 * beside the periods that follow; the main stream waits for it only when its output is due, so
 * the longest period costs about what the common one does (tools/nupc_latency.py).
 *
 * Filters are single-input single-output impulse responses (a crossbar is one call per pair).
 * Raw I/O buffers hold interleaved frames (dai.c's interleaved layout): all channels of a side
 * share sample_spacing and bytes.
 */
#ifndef BFHIP_NUPC_H
#define BFHIP_NUPC_H

#include "bfhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bfhip_nupc bfhip_nupc;

const char *bfhip_nupc_last_error(void);
bfhip_nupc *bfhip_nupc_create(int device, int realsize, int n_in, int n_out, int n_segments,
                              const int seg_length[], const int seg_blocks[]);
void bfhip_nupc_destroy(bfhip_nupc *n);
long bfhip_nupc_taps(const bfhip_nupc *n);        /* taps covered by the schedule */
int bfhip_nupc_latency(const bfhip_nupc *n);      /* I/O block size in frames = seg_length[0] */
int bfhip_nupc_set_format(bfhip_nupc *n, int io, int channel, const bfhip_format *bf);
int bfhip_nupc_set_safety_limit(bfhip_nupc *n, double limit);
int bfhip_nupc_add_filter(bfhip_nupc *n, int in_channel, int out_channel, const void *taps,
                          long n_taps, double in_scale, double out_scale);
int bfhip_nupc_finalize(bfhip_nupc *n);
/* one I/O block of seg_length[0] frames; host buffers, synchronous; returns status bits
   (BFHIP_ST_*) or a negative error */
int bfhip_nupc_block(bfhip_nupc *n, const void *rawin, void *rawout, bfhip_overflow overflow[]);
/* device-resident buffers, asynchronous on the convolver's stream */
int bfhip_nupc_block_dev(bfhip_nupc *n, const void *rawin_dev, void *rawout_dev);
/* waits for the periods handed in so far and returns the status bits collected since the last
   call (background segment blocks that are not due yet keep running) */
int bfhip_nupc_sync(bfhip_nupc *n);
int bfhip_nupc_get_overflow(bfhip_nupc *n, int out_channel, bfhip_overflow *of);

#ifdef __cplusplus
}
#endif
#endif
