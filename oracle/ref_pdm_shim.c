/* ref_pdm_shim.c -- builds the REAL reference header stm32f103/pdm.h (which is
 * self-contained: it includes only <stdint.h>) from where it lies under
 * /root/reference, and gives its static-inline functions external names so
 * tests can call them through ctypes.  No reference source is copied: the
 * header is pulled in by -I at build time (oracle/Makefile) and the result
 * goes to oracle/_ref/ (git-ignored, travels to the GPU box as a .so).
 * ORACLE / test infrastructure only. */
#include "pdm.h"
/* pdm.h ends each function body with a stray attribute that binds to the next
 * declaration (pdm.h:24,40,57,77); give the last one something harmless. */
struct ref_pdm_attribute_sink;

uint32_t ref_pdm1_update(uint32_t *s, uint32_t input, uint32_t out_shift) {
    return pdm1_update((struct pdm1 *)s, input, out_shift);
}
uint32_t ref_pdm2_update(uint32_t *s, uint32_t input, uint32_t out_shift, uint32_t dither) {
    return pdm2_update((struct pdm2 *)s, input, out_shift, dither);
}
uint32_t ref_pdm3_update(uint32_t *s, uint32_t input, uint32_t out_shift, uint32_t dither) {
    return pdm3_update((struct pdm3 *)s, input, out_shift, dither);
}
uint32_t ref_pdm4_update(uint32_t *s, uint32_t input, uint32_t out_shift, uint32_t dither) {
    return pdm4_update((struct pdm4 *)s, input, out_shift, dither);
}
