/* Where does a synchronous smx_bank_run spend its time?  (GPU box; links the product library.)
 *   gcc -O2 -I include -o /tmp/sync_path tools/ubench/sync_path.c -L synth_tools_amd -lsynth_mi355x -Wl,-rpath,$PWD/synth_tools_amd -Wl,-rpath,/opt/rocm/lib
 * Prints, per bank size, the mean host time of smx_bank_run_async (the launch), of smx_bank_fetch (publish kernel +
 * poll, or copy + stream sync with SMX_NO_PUBLISH=1) and of the whole smx_bank_run, 64-frame blocks. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "synth_mi355x.h"

static double now_us(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }

int main(void)
{
    const uint32_t sizes[] = {64, 65536, 1u << 20, 1u << 24};
    for (unsigned s = 0; s < sizeof sizes / sizeof sizes[0]; s++) {
        const uint32_t n = sizes[s];
        smx_bank *b = smx_bank_create(n, 0);
        if (!b) { fprintf(stderr, "%s\n", smx_last_error()); return 1; }
        uint32_t *inc = malloc((size_t)n * 4), *st = malloc((size_t)n * 4);
        for (uint32_t v = 0; v < n; v++) { inc[v] = note_to_inc(21 + (int)(v % 88)); st[v] = v * 2654435761u; }
        if (smx_bank_load(b, inc, st)) return 1;
        float vec[64];
        for (int i = 0; i < 300; i++) smx_bank_run(b, vec, NULL, 64);
        double ta = 0, tf = 0, tr = 0;
        const int reps = 2000;
        for (int i = 0; i < reps; i++) {
            const double t0 = now_us();
            smx_bank_run_async(b, 64);
            const double t1 = now_us();
            smx_bank_fetch(b, vec, NULL, 64);
            const double t2 = now_us();
            ta += t1 - t0; tf += t2 - t1;
        }
        for (int i = 0; i < reps; i++) { const double t0 = now_us(); smx_bank_run(b, vec, NULL, 64); tr += now_us() - t0; }
        printf("%9u voices x 64 frames: run_async %5.2f us  fetch %5.2f us  | smx_bank_run %5.2f us per block\n", n, ta / reps, tf / reps, tr / reps);
        smx_bank_destroy(b);
        free(inc); free(st);
    }
    return 0;
}
