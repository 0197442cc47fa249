/* test_synth_abi.c -- a reference-style host test (plain C, ASSERT-based, exit status 0 on
 * success; the harness pattern of the reference's rules.mk:382-386 "run the binary") against
 * libsynth_mi355x.so.  Needs a GPU.  Known answers: SURVEY.md Appendix A.2 (recorded from the
 * reference) and the comment KAT of stm32f103/mod_pdm.c:43-47. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "synth_mi355x.h"

#define LOG(...) fprintf(stderr, __VA_ARGS__)
#define ASSERT(x) do { if (!(x)) { LOG("%s:%d: ASSERT(%s) failed: %s\n", __FILE__, __LINE__, #x, smx_last_error()); exit(1); } } while (0)

static uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static void test_dropin_chord(void) {
    /* synth_init; note_on 69,72,76; synth_run(...,8) */
    static const uint32_t want[8] = {0x00000000, 0x3b0a742f, 0x3b8a7430, 0x3bcfae48,
                                     0x3c0a7430, 0x3c2d113c, 0x3c4fae48, 0x3c724b54};
    struct synth s;
    float vec[8];
    synth_init(&s);
    synth_note_on(&s, 69); synth_note_on(&s, 72); synth_note_on(&s, 76);
    synth_run(&s, vec, 8);
    for (int i = 0; i < 8; i++) ASSERT(bits_of(vec[i]) == want[i]);
    ASSERT(s.voice[0].note_inc == 0x0258bf25 && s.voice[0].note_state == 0x12c5f928);
    ASSERT(note_to_inc(69) == 39370533);
    /* stray note-off silences voice 0; a full bank steals voice 0 (linux/synth.c:145-165) */
    synth_note_off(&s, 100);
    ASSERT(s.voice[0].note_inc == 0);
    for (int n = 0; n < 70; n++) synth_note_on(&s, n);
    ASSERT(s.note2voice[69] == 0 && voice_alloc(&s) == 0);
}

static void test_bank_equals_dropin(void) {
    /* the same three notes on a 100000-voice bank give the same samples */
    struct synth s;
    float a[64], b[64];
    synth_init(&s);
    smx_bank *bank = smx_bank_create(100000, 0);
    ASSERT(bank);
    const int notes[3] = {69, 72, 76};
    for (int i = 0; i < 3; i++) { synth_note_on(&s, notes[i]); ASSERT(0 == smx_bank_note_on(bank, notes[i])); }
    for (int blk = 0; blk < 5; blk++) {
        synth_run(&s, a, 64);
        ASSERT(0 == smx_bank_run(bank, b, NULL, 64));
        ASSERT(0 == memcmp(a, b, sizeof a));
    }
    ASSERT(0 == smx_bank_note_off(bank, 72)); synth_note_off(&s, 72);
    synth_run(&s, a, 64);
    ASSERT(0 == smx_bank_run(bank, b, NULL, 64));
    ASSERT(0 == memcmp(a, b, sizeof a));
    ASSERT(smx_bank_voices(bank) == 100000);
    smx_bank_destroy(bank);
}

static void test_pdm_comment_kat(void) {
    /* 3-bit accumulator, X = 3: C = 1 0 0 1 0 0 1 0 1 starting from A = 5 (mod_pdm.c:43-47) */
    static const uint32_t want[9] = {1, 0, 0, 1, 0, 0, 1, 0, 1};
    smx_pdm *p = smx_pdm_create(1, 0);
    ASSERT(p);
    uint32_t sp = 3u << 29, ac = 5u << 29, bits[9];
    ASSERT(0 == smx_pdm_load(p, &sp, &ac));
    ASSERT(0 == smx_pdm_tick_n(p, 9, NULL, bits));
    for (int i = 0; i < 9; i++) ASSERT(bits[i] == want[i]);
    ASSERT(smx_pdm_set_setpoint(p, 1, 0) == SMX_E_RANGE);       /* mod_synth.c:107 */
    smx_pdm_destroy(p);
}

static void test_firmware_packet(void) {
    /* the reference's own example: bp2 ! {send_packet, <<16#FFF50002:32, 100:32, 1:32>>} */
    static const uint8_t mode_on[12] = {0xFF, 0xF5, 0x00, 0x02, 0, 0, 0, 100, 0, 0, 0, 1};
    static const uint8_t mode_off[12] = {0xFF, 0xF5, 0x00, 0x02, 0, 0, 0, 100, 0, 0, 0, 0};
    smx_fw *fw = smx_fw_create(3, 1, 0);
    ASSERT(fw);
    ASSERT(0 == smx_fw_handle_packet(fw, mode_off, sizeof mode_off) && !smx_fw_running(fw));
    ASSERT(0 == smx_fw_handle_packet(fw, mode_on, sizeof mode_on) && smx_fw_running(fw));
    uint8_t duty[300];
    ASSERT(100 == smx_fw_tick_n(fw, 100, NULL, duty));
    smx_fw_destroy(fw);
}

int main(void) {
    if (smx_device_count() < 1) { LOG("test_synth_abi.c: no GPU\n"); return 2; }
    test_dropin_chord();
    test_bank_equals_dropin();
    test_pdm_comment_kat();
    test_firmware_packet();
    LOG("test_synth_abi.c\n");          /* the reference's tests log their own name (linux/test_pdm.c:15) */
    return 0;
}
