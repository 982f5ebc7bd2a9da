// host_ops_driver.cpp -- drives the pure-host half of the convolver.h boundary (csrc/host_ops.cpp)
// from a program of its own, so that it can be compiled with AddressSanitizer / UBSan (the GPU
// pool offers no device sanitizer; the host code can have one).  Built and run by
// tests/test_host_ops.py::test_host_ops_under_address_and_ub_sanitizers; exit code 0 = clean.
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bfhip_convolver.h"

extern "C" void bfhip_coeff_mark_dirty(const void *cbuf);
extern "C" unsigned long long bfhip_coeff_dirty_sequence(void);

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "host_ops_driver: %s failed (line %d)\n", #c, __LINE__); return 1; } } while (0)

template <typename T> static int run(int L) {
    const int rs = (int)sizeof(T);
    CHECK(convolver_init("unused-wisdom", L, rs) == 1);
    CHECK(convolver_cbufsize() == 2 * L * rs);
    const int n_taps = L + 5 > 256 ? L + 5 : 256;          // the td filters below read up to 199 of them
    std::vector<T> taps(n_taps), dest(2 * L), dest2(2 * L);
    for (int i = 0; i < n_taps; i++) taps[i] = (T)std::sin(0.37 * i) / (T)(1 + i % 7);
    // every tap count around the partition boundary, with and without a destination
    for (int n : {0, 1, L / 2 + 1, L - 1, L, L + 5}) {
        CHECK(convolver_coeffs2cbuf(taps.data(), n, 0.5, dest.data()) == dest.data());
        void *own = convolver_coeffs2cbuf(taps.data(), n, 0.5, NULL);
        CHECK(own != NULL && memcmp(own, dest.data(), dest.size() * sizeof(T)) == 0);
    }
    taps[3] = (T)NAN;
    CHECK(convolver_coeffs2cbuf(taps.data(), L, 1.0, dest.data()) == NULL);
    taps[3] = (T)0.25;
    convolver_runtime_coeffs2cbuf(taps.data(), dest.data());
    CHECK(convolver_coeffs2cbuf(taps.data(), L, 1.0, dest2.data()) == dest2.data());
    for (int i = 0; i < 2 * L; i++) CHECK(std::fabs((double)dest[i] - (double)dest2[i]) <= 1e-6 * (1 + std::fabs((double)dest2[i])));
    void *bufs[2] = {dest.data(), dest2.data()};
    CHECK(convolver_verify_cbuf(bufs, 2) == 1);
    char path[] = "/tmp/bfhip_dump_XXXXXX";
    const int fd = mkstemp(path);
    CHECK(fd >= 0);
    close(fd);
    convolver_debug_dump_cbuf(path, bufs, 2);
    FILE *f = fopen(path, "rt");
    CHECK(f != NULL);
    int lines = 0;
    double v;
    while (fscanf(f, "%lf", &v) == 1) {
        if (lines < L) CHECK(std::fabs(v - (double)taps[lines]) <= 1e-4);
        lines++;
    }
    fclose(f);
    unlink(path);
    CHECK(lines == 2 * L);
    // fft plans of every order, in place and out of place, forward then back
    for (int order = 1; order <= 12; order++) {
        const int n = 1 << order;
        std::vector<T> x(n), y(n), z(n);
        for (int i = 0; i < n; i++) x[i] = (T)std::cos(0.11 * i * i);
        bfhip_fftplan_execute(convolver_fftplan(order, 0, 0), x.data(), y.data());
        z = y;
        bfhip_fftplan_execute(convolver_fftplan(order, 1, 1), z.data(), z.data());
        for (int i = 0; i < n; i++) CHECK(std::fabs((double)z[i] / n - (double)x[i]) <= (rs == 4 ? 2e-5 : 1e-12));
    }
    // td filters of awkward lengths
    for (int n : {1, 2, 3, 31, 32, 33, 199}) {
        CHECK(convolver_td_block_length(n) >= n);
        CHECK(convolver_td_new(taps.data(), n) != NULL);
    }
    CHECK(convolver_td_new(taps.data(), 0) == NULL);
    return 0;
}

int main() {
    for (int L : {4, 64, 1024}) {
        if (run<float>(L) != 0 || run<double>(L) != 0) return 1;
    }
    // change notices: many addresses (collisions in the open-addressing table), from a child too
    CHECK(convolver_init(NULL, 64, 4) == 1);
    const unsigned long long s0 = bfhip_coeff_dirty_sequence();
    std::vector<float> arena(64 * 20000);
    for (int i = 0; i < 20000; i++) bfhip_coeff_mark_dirty(&arena[(size_t)i * 64]);      // more than the table holds
    CHECK(bfhip_coeff_dirty_sequence() == s0 + 20000);
    const pid_t pid = fork();
    if (pid == 0) {
        bfhip_coeff_mark_dirty(&arena[0]);
        _exit(0);
    }
    int st = 0;
    CHECK(waitpid(pid, &st, 0) == pid && WIFEXITED(st) && WEXITSTATUS(st) == 0);
    CHECK(bfhip_coeff_dirty_sequence() == s0 + 20001);
    printf("host_ops_driver: clean\n");
    return 0;
}
