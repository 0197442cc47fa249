/* fw_host.c -> host/fw.dynamic.host.elf
 *
 * The firmware's communication loop (stm32f103/synth.c:27-42 handle_tag +
 * mod_synth.c:89-137 synth_handle_tag_u32), hosted as an Erlang-style port
 * program: {packet,4} frames on stdin, as erl/jack_client.erl:63-68 opens its ports
 * and linux/clock.c:229-261 reads them.  TAG_U32 frames are the firmware's commands
 * (MODE / SETPOINT / MEASURE / parameters).  One extension tag drives time, which on
 * the MCU is the timer interrupt: TAG_STREAM (0xFFFB, erl/jack_client.erl:28) with
 * stream id 1 and a big-endian u32 tick count runs that many PDM ISR ticks on the GPU
 * and answers with one {packet,4} frame <<0xFFFB:16, 1:16, duty bytes (tick-major)>>.
 * EOF on stdin ends the program with status 1 (linux/synth.c:305-310 convention).
 *
 *   fw.dynamic.host.elf [n_channels [n_oscillators]]      defaults 3 1 (mod_pdm_pwm.c:42-43)
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "synth_mi355x.h"

#define LOG(...) fprintf(stderr, __VA_ARGS__)
#define ASSERT(x) do { if (!(x)) { LOG("%s:%d: ASSERT(%s) failed\n", __FILE__, __LINE__, #x); exit(1); } } while (0)

static void read_fixed(int fd, uint8_t *buf, size_t n) {
    while (n) {
        ssize_t r = read(fd, buf, n);
        if (r == 0) exit(1);
        if (r < 0) { if (errno == EINTR) continue; exit(1); }
        buf += r; n -= (size_t)r;
    }
}
static void write_fixed(int fd, const uint8_t *buf, size_t n) {
    while (n) {
        ssize_t r = write(fd, buf, n);
        if (r < 0) { if (errno == EINTR) continue; exit(1); }
        buf += r; n -= (size_t)r;
    }
}
static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static void put_be32(uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

int main(int argc, char **argv) {
    uint32_t nch = argc > 1 ? (uint32_t)strtoul(argv[1], NULL, 0) : 3;
    uint32_t nosc = argc > 2 ? (uint32_t)strtoul(argv[2], NULL, 0) : 1;
    smx_fw *fw = smx_fw_create(nch, nosc, 0);
    if (!fw) { LOG("fw: %s\n", smx_last_error()); exit(1); }
    for (;;) {
        uint8_t hdr[4];
        read_fixed(0, hdr, 4);
        uint32_t nb = be32(hdr);
        uint8_t *buf = malloc(nb ? nb : 1);
        ASSERT(buf);
        read_fixed(0, buf, nb);
        if (nb >= 8 && buf[0] == 0xFF && buf[1] == 0xFB && buf[2] == 0 && buf[3] == 1) {
            uint32_t nt = be32(buf + 4);
            size_t out_n = (size_t)nt * nch;
            uint8_t *out = malloc(8 + out_n);
            ASSERT(out);
            int ran = smx_fw_tick_n(fw, nt, NULL, out + 8);
            ASSERT(ran >= 0);
            if (ran == 0) out_n = 0;                       /* stopped: no ISR ran */
            put_be32(out, (uint32_t)(4 + out_n));
            out[4] = 0xFF; out[5] = 0xFB; out[6] = 0; out[7] = 1;
            write_fixed(1, out, 8 + out_n);
            free(out);
        } else {
            smx_fw_handle_packet(fw, buf, nb);
        }
        free(buf);
    }
    return 0;
}
