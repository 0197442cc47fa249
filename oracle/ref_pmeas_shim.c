/* ref_pmeas_shim.c -- NOT a translation unit of its own: oracle/Makefile streams
 * `struct pmeas`, `struct pmeas_state` and `pmeas_update` out of the reference's
 * stm32f103/pmeas.h (:4-28, :64-108) into gcc's stdin and appends this fragment,
 * which only gives the reference's static-inline function an external name and
 * tells the test the layout of the reference's struct.  No reference source is
 * copied into the repo; the result goes to oracle/_ref/ (git-ignored .so).
 * ORACLE / test infrastructure only. */
#include <stddef.h>

void ref_pmeas_update(struct pmeas_state *p, uint32_t cc) {
    pmeas_update(p, cc);
}
/* offsets of the fields the tests read: log_max, write, read, meas[0].avg, meas[0].num,
 * meas[1].avg, meas[1].num, num, accu, last_cc; then sizeof */
void ref_pmeas_layout(uint32_t out[11]) {
    out[0] = offsetof(struct pmeas_state, log_max);
    out[1] = offsetof(struct pmeas_state, write);
    out[2] = offsetof(struct pmeas_state, read);
    out[3] = offsetof(struct pmeas_state, meas[0].avg);
    out[4] = offsetof(struct pmeas_state, meas[0].num);
    out[5] = offsetof(struct pmeas_state, meas[1].avg);
    out[6] = offsetof(struct pmeas_state, meas[1].num);
    out[7] = offsetof(struct pmeas_state, num);
    out[8] = offsetof(struct pmeas_state, accu);
    out[9] = offsetof(struct pmeas_state, last_cc);
    out[10] = sizeof(struct pmeas_state);
}
