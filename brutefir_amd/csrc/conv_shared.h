// conv_shared.h -- what the two halves of the convolver.h boundary share inside libbfhip.so:
// host_ops.cpp (pure host: runs in bfconf's parent before the fork and in module processes,
// compiled with g++, cannot call HIP) and convolver_abi.hip (the per-call device ops).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

extern "C" {

// convolver_init()'s arguments (fftw_convolver.c:36-49 keeps the same in file statics)
struct bfhip_conv_globals {
    int L, rs, log2L, inited;
};
extern struct bfhip_conv_globals bfhip_conv_g;

// bf_exit()-style failure of a call that cannot return an error: records the code, calls the
// installed handler or prints and exit(1)s (bfhip_convolver_set_fatal_handler)
void bfhip_conv_fatal(int code, const char *message);

// Coefficient change notices between processes (bflogic_eq -> filter process), see bfhip.h.
// Slots live in a MAP_SHARED mapping created by convolver_init() in the parent, before fork().
struct bfhip_dirty_slot { uintptr_t addr; uint64_t gen; };
#define BFHIP_DIRTY_SLOTS 8192
struct bfhip_dirty_table {
    uint64_t seq;                                   // bumped by every notice
    uint64_t lost;                                  // notices that found no slot: watchers re-read everything
    struct bfhip_dirty_slot slot[BFHIP_DIRTY_SLOTS];
};
struct bfhip_dirty_table *bfhip_dirty_table_get(void);    // NULL before convolver_init()
uint64_t bfhip_dirty_generation(const void *cbuf);         // 0 = never marked
uint64_t bfhip_dirty_lost(void);                           // notices the table had no room for

}  // extern "C"

// td_conv_t (fftw_convolver.c:682-687): the filter's spectrum is computed on the host at
// start-up (delay.c builds its filters in the parent); the device copy convolver_td_convolve
// works with is made on first use in the process that convolves
struct _td_conv_t_ {
    void *h_coeffs;        // [2 * blocklen] reals, halfcomplex, already / (2 * blocklen)
    int blocklen;
    void *d_coeffs;        // device copy, owned by process d_pid
    pid_t d_pid;
};
