/* ref_pwmosc_shim.c -- NOT a translation unit of its own: oracle/Makefile streams the 17 lines
 * `#define OSC_HARD_SYNC` ... end of `pwm_update` out of the reference's stm32f103/mod_pdm.c
 * (:159-175: OSC_HARD_SYNC, pwm_phase, pwm_speed, PHASE_MASK, CONTROL_DIV, control_div_count,
 * pwm_update -- plain C over <stdint.h>; the rest of that file needs ARM asm and libopencm3 and
 * is NOT built) into gcc's stdin and appends this fragment, which only gives the reference's
 * static-inline function and its macro external names and drives them tick by tick.  No
 * reference source is copied into the repo; the result goes to oracle/_ref/ (git-ignored .so).
 * ORACLE / test infrastructure only. */

uint32_t ref_pwm_update(void) { return pwm_update(); }
void ref_osc_hard_sync(void) { OSC_HARD_SYNC(); }
uint32_t ref_pwm_get_phase(void) { return pwm_phase; }
uint32_t ref_pwm_get_speed(void) { return pwm_speed; }          /* the reference's default: 256 * 13 */
void ref_pwm_set(uint32_t phase, uint32_t speed) { pwm_phase = phase; pwm_speed = speed; }
uint32_t ref_pwm_control_div(void) { return CONTROL_DIV; }

/* n ticks of the ISR's `uint32_t val = pwm_update()` (mod_pdm.c:182); sync[t] != 0: the oscillator
 * ISR's OSC_HARD_SYNC() (mod_osc.c:60-62) lands before tick t.  duty[t] = val as the 8-bit compare value. */
void ref_pwmosc_run(uint32_t n, const uint8_t *sync, uint8_t *duty, uint32_t *phase_after) {
    for (uint32_t t = 0; t < n; t++) {
        if (sync && sync[t]) OSC_HARD_SYNC();
        uint32_t v = pwm_update();
        if (duty) duty[t] = (uint8_t)v;
        if (phase_after) phase_after[t] = pwm_phase;
    }
}
