/* synth_host.c -> host/synth.dynamic.host.elf
 *
 * The JACK client of linux/synth.c:214-312, rewritten as host glue over
 * libsynth_mi355x.so: same client name ("synth"), same ports ("midi_in",
 * "audio_out"), same callback order (MIDI, then audio: linux/synth.c:277-282),
 * same life cycle (open, register, set callback, mlockall, activate,
 * synth_init, then block on a 4-byte stdin read and exit(1):
 * linux/synth.c:285-312), so that erl/jack_client.erl:63-82 can spawn it
 * under the same file name with {packet,4} stdio.
 *
 * Plain C.  libjack is bound at run time (dlopen) against the public JACK
 * ABI, because this image has no JACK headers; with --fake-jack the same
 * process() callback is driven by a scripted block loop instead (the
 * reference's "stub the environment, run the real code" test pattern,
 * linux/test_bl_midi.c:5-47).
 *
 *   synth.dynamic.host.elf                       real JACK, 64-voice drop-in path
 *   SYNTH_VOICES=1048576 synth.dynamic.host.elf  real JACK, N-voice bank path
 *   SYNTH_FILL=1                                  the bank starts with every voice sounding
 *   SYNTH_PIPELINE=1 ...                         bank path returns block k-1 while k computes
 *   SYNTH_RANKS=8 SYNTH_VOICES=8388608 ...       the bank sharded over 8 GPUs: 7 helper processes (forked before any
 *                                                GPU call) hold the other shards; this process stays the only JACK
 *                                                client, forwards every block's MIDI events and frame count to the
 *                                                helpers over pipes, and the library sums the buses (RCCL).
 *                                                SYNTH_DEVICES=0,1,... maps ranks to HIP devices (default: rank r ->
 *                                                device r)
 *   SYNTH_FAKE_PERIOD_US=1333 ... --fake-jack    pace the scripted callbacks like a sound card
 *   synth.dynamic.host.elf --fake-jack NBLOCKS NFRAMES EVENTS.bin OUT.f32
 *       EVENTS.bin: records {u32 block; u8 size; u8 bytes[3]} delivered to
 *       midi_in at the start of that block; OUT.f32: NBLOCKS*NFRAMES floats.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "synth_mi355x.h"

#define LOG(...) fprintf(stderr, __VA_ARGS__)
/* linux/erl_tools_system.h:15,24-27 */
#define ASSERT(x) do { if (!(x)) { LOG("%s:%d: ASSERT(%s) failed\n", __FILE__, __LINE__, #x); exit(1); } } while (0)

/* ---- the public JACK ABI, as far as linux/synth.c uses it ------------------ */
typedef uint32_t jack_nframes_t;
typedef float jack_default_audio_sample_t;
typedef struct _jack_client jack_client_t;
typedef struct _jack_port jack_port_t;
typedef struct { jack_nframes_t time; size_t size; uint8_t *buffer; } jack_midi_event_t;
typedef int (*JackProcessCallback)(jack_nframes_t nframes, void *arg);
#define JACK_DEFAULT_AUDIO_TYPE "32 bit float mono audio"
#define JACK_DEFAULT_MIDI_TYPE "8 bit raw midi"
enum { JackPortIsInput = 0x1, JackPortIsOutput = 0x2 };
enum { JackNullOption = 0x00 };

static struct {
    jack_client_t *(*client_open)(const char *, int, int *, ...);
    jack_port_t *(*port_register)(jack_client_t *, const char *, const char *, unsigned long, unsigned long);
    void *(*port_get_buffer)(jack_port_t *, jack_nframes_t);
    int (*set_process_callback)(jack_client_t *, JackProcessCallback, void *);
    int (*activate)(jack_client_t *);
    uint32_t (*midi_get_event_count)(void *);
    int (*midi_event_get)(jack_midi_event_t *, void *, uint32_t);
} jack;

/* ---- fake JACK: one MIDI buffer and one audio buffer per block ------------- */
struct fake_event { uint32_t block; uint8_t size; uint8_t bytes[3]; };
static struct {
    int on;
    struct fake_event *ev; size_t nev, cursor;
    uint32_t block;
    jack_midi_event_t cur[1024]; uint32_t ncur;
    float *audio;
} fake;
static jack_port_t *const FAKE_MIDI_PORT = (jack_port_t *)1, *const FAKE_AUDIO_PORT = (jack_port_t *)2;

static void *fake_port_get_buffer(jack_port_t *p, jack_nframes_t n) { (void)n; return p == FAKE_AUDIO_PORT ? (void *)fake.audio : (void *)&fake; }
static uint32_t fake_midi_get_event_count(void *b) { (void)b; return fake.ncur; }
static int fake_midi_event_get(jack_midi_event_t *e, void *b, uint32_t i) { (void)b; if (i >= fake.ncur) return -1; *e = fake.cur[i]; return 0; }

/* ---- SYNTH: either the reference's struct synth or an N-voice bank --------- */
static struct synth synth;               /* linux/synth.c:208 */
static smx_bank *bank;                   /* SYNTH_VOICES > 64 */

/* SYNTH_RANKS > 1: this process is rank 0 and the only JACK client; helper[r] is the write end of rank r's pipe. */
static int n_ranks = 1, my_rank = 0;
static int helper[8];
struct rank_cmd { uint32_t nframes, nev; };          /* then 3 * nev bytes of events; nframes 0: events only */

/* A helper that dies must end the client, loudly, wherever rank 0 is at that moment -- also inside an RCCL sum that
 * the dead rank will never join (no watchdog could return from there).  SIGCHLD does that: the handler reaps, says so
 * and leaves with status 1 (async-signal-safe calls only); a supervisor (jack_client.erl sees exit_status) starts a
 * fresh process -- never a re-exec from a process that has touched the GPU.  SIGPIPE is ignored so that a write to a
 * dead helper's pipe reports EPIPE here instead of killing the client silently. */
static void on_sigchld(int sig) {
    (void)sig;
    int st;
    if (waitpid(-1, &st, WNOHANG) > 0) {
        static const char msg[] = "synth: a helper rank is gone\n";
        if (write(2, msg, sizeof msg - 1) < 0) { /* nothing to do about it */ }
        _exit(1);
    }
}
static void write_all(int fd, const void *buf, size_t n) {
    const uint8_t *p = buf;
    while (n) {
        ssize_t w = write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; LOG("synth: a helper rank is gone\n"); exit(1); }
        p += w; n -= (size_t)w;
    }
}
static int read_all(int fd, void *buf, size_t n) {      /* 0 on EOF */
    uint8_t *p = buf;
    while (n) {
        ssize_t r = read(fd, p, n);
        if (r == 0) return 0;
        if (r < 0) { if (errno == EINTR) continue; return 0; }
        p += r; n -= (size_t)r;
    }
    return 1;
}
static void to_helpers(uint32_t nframes, const uint8_t *ev3, uint32_t nev) {
    struct rank_cmd c = { nframes, nev };
    for (int r = 1; r < n_ranks; r++) {
        write_all(helper[r], &c, sizeof c);
        if (nev) write_all(helper[r], ev3, 3u * nev);
    }
}
static int device_of(int rank) {
    const char *d = getenv("SYNTH_DEVICES");
    for (int r = 0; d && r < rank; r++) { d = strchr(d, ','); if (d) d++; }
    return d ? atoi(d) : rank;
}

/* SYNTH_FILL=1: start with every voice of the bank sounding (notes 21..108 spread over the voices,
 * scattered phases) instead of silence -- what a latency measurement of a full bank needs. */
static void bank_fill(smx_bank *b, uint32_t first, uint32_t n) {
    uint32_t *inc = malloc((size_t)n * 4), *st = malloc((size_t)n * 4);
    ASSERT(inc && st);
    for (uint32_t v = 0; v < n; v++) {
        uint32_t h = (first + v) * 2654435761u;          /* by GLOBAL voice number: the same bank however it is sharded */
        /* SYNTH_FILL=worst: every voice wraps 12 times per 64 frames (increment 12/64 of 2^32), the
           bank on which locating the wraps costs most; anything else: piano range */
        const char *fill = getenv("SYNTH_FILL");
        inc[v] = (fill && !strcmp(fill, "worst")) ? (0x30000000u | (h & 0xFFFFu))
                                                  : note_to_inc(21 + (int)((h >> 12) % 88u));
        st[v] = h * 40503u + 12345u;
    }
    ASSERT(0 == smx_bank_load(b, inc, st));
    free(inc); free(st);
}

/* The N-voice bank of this rank (SYNTH_VOICES > 64): its shard of the voices, the communicator, the options. */
static void setup_bank(uint32_t voices, const uint8_t *id) {
    if (voices <= 64) return;
    const uint32_t per = voices / (uint32_t)n_ranks;
    ASSERT(n_ranks == 1 || (voices % (uint32_t)n_ranks == 0 && per % 64 == 0));
    ASSERT((bank = smx_bank_create(per, device_of(my_rank))));
    if (n_ranks > 1) {
        ASSERT(0 == smx_bank_comm_init(bank, my_rank, n_ranks, id));
        ASSERT(0 == smx_bank_shard(bank, (uint32_t)my_rank * per, voices));    /* the allocator spans all ranks */
    }
    if (getenv("SYNTH_PIPELINE")) ASSERT(0 == smx_bank_set_block_mode(bank, SMX_BLOCK_PIPELINED));
    if (getenv("SYNTH_FORM_STEPPING")) ASSERT(0 == smx_bank_set_block_form(bank, SMX_FORM_STEPPING));
    if (getenv("SYNTH_FILL")) bank_fill(bank, (uint32_t)my_rank * per, per);
}

/* A helper rank: no JACK, no stdin -- it does what rank 0 tells it, block by block, and leaves when the pipe closes. */
static void helper_loop(int fd, uint32_t voices) {
    uint8_t id[SMX_UNIQUE_ID_BYTES];
    if (!read_all(fd, id, sizeof id)) _exit(0);
    setup_bank(voices, id);
    static uint8_t ev[3 * 4096];
    struct rank_cmd c;
    while (read_all(fd, &c, sizeof c)) {
        ASSERT(c.nev <= 4096);
        if (c.nev) { if (!read_all(fd, ev, 3u * c.nev)) break; ASSERT(0 == smx_bank_midi_events(bank, ev, c.nev)); }
        if (c.nframes) ASSERT(0 == smx_bank_run(bank, NULL, NULL, (int)c.nframes));
    }
    /* rank 0 closed the pipe (or is gone): release the shard and its communicator; should a peer that has already
       left keep the teardown waiting, SIGALRM's default action ends this process after 5 s */
    alarm(5);
    smx_bank_destroy(bank);
    _exit(0);
}
static jack_port_t *midi_in, *audio_out; /* linux/synth.c:214-221 */

static inline void process_midi(jack_nframes_t nframes) {        /* linux/synth.c:227-260 */
    void *midi_in_buf = jack.port_get_buffer(midi_in, nframes);
    jack_nframes_t n = jack.midi_get_event_count(midi_in_buf);
    static uint8_t batch[3 * 4096];          /* bank mode: the block's 3-byte events, one call */
    size_t nbatch = 0;
    for (jack_nframes_t i = 0; i < n; i++) {
        jack_midi_event_t event;
        jack.midi_event_get(&event, midi_in_buf, i);
        const uint8_t *msg = event.buffer;
        if (!bank) { synth_midi_event(&synth, msg, event.size); continue; }
        if (event.size != 3) continue;           /* linux/synth.c:236: only 3-byte events */
        if (nbatch == 4096) { to_helpers(0, batch, 4096); ASSERT(0 == smx_bank_midi_events(bank, batch, nbatch)); nbatch = 0; }
        memcpy(batch + 3 * nbatch++, msg, 3);
    }
    if (nbatch) { to_helpers(0, batch, (uint32_t)nbatch); ASSERT(0 == smx_bank_midi_events(bank, batch, nbatch)); }
}
static inline void process_audio(jack_nframes_t nframes) {       /* linux/synth.c:261-276 */
    jack_default_audio_sample_t *dst = jack.port_get_buffer(audio_out, nframes);
    if (bank) { to_helpers(nframes, NULL, 0); ASSERT(0 == smx_bank_run(bank, dst, NULL, (int)nframes)); }
    else      synth_run(&synth, dst, (int)nframes);
}
static int process(jack_nframes_t nframes, void *arg) {          /* linux/synth.c:277-282 */
    (void)arg;
    /* Order is important. */
    process_midi(nframes);
    process_audio(nframes);
    return 0;
}

static void bind_jack(void) {
    void *h = dlopen("libjack.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { LOG("synth: cannot load libjack.so.0 (%s); use --fake-jack\n", dlerror()); exit(1); }
#define SYM(field, name) ASSERT((*(void **)&jack.field = dlsym(h, name)))
    SYM(client_open, "jack_client_open");
    SYM(port_register, "jack_port_register");
    SYM(port_get_buffer, "jack_port_get_buffer");
    SYM(set_process_callback, "jack_set_process_callback");
    SYM(activate, "jack_activate");
    SYM(midi_get_event_count, "jack_midi_get_event_count");
    SYM(midi_event_get, "jack_midi_event_get");
#undef SYM
}

static void read_fixed(int fd, uint8_t *buf, size_t n) {         /* assert_read, linux/synth.c:308 */
    while (n) {
        ssize_t r = read(fd, buf, n);
        if (r == 0) exit(1);                                     /* EOF: the port was closed */
        if (r < 0) { if (errno == EINTR) continue; exit(1); }
        buf += r; n -= (size_t)r;
    }
}

int main(int argc, char **argv) {
    const char *voices_env = getenv("SYNTH_VOICES");
    uint32_t voices = voices_env ? (uint32_t)strtoul(voices_env, NULL, 0) : 64;
    uint8_t id[SMX_UNIQUE_ID_BYTES] = {0};
    if (getenv("SYNTH_RANKS") && voices > 64) {
        /* fork the helper ranks FIRST: no process may have touched the GPU when it forks */
        n_ranks = atoi(getenv("SYNTH_RANKS"));
        ASSERT(n_ranks >= 1 && n_ranks <= 8);
        if (n_ranks > 1) { signal(SIGPIPE, SIG_IGN); signal(SIGCHLD, on_sigchld); }
        for (int r = 1; r < n_ranks; r++) {
            int fd[2];
            ASSERT(0 == pipe(fd));
            pid_t pid = fork();
            ASSERT(pid >= 0);
            if (pid == 0) {
                close(fd[1]);
                for (int k = 1; k < r; k++) close(helper[k]);      /* the other helpers' pipes are not ours */
                close(0);
                signal(SIGCHLD, SIG_DFL);
                my_rank = r;
                helper_loop(fd[0], voices);
            }
            close(fd[0]);
            helper[r] = fd[1];
        }
        if (n_ranks > 1) {
            ASSERT(0 == smx_comm_unique_id(id));
            for (int r = 1; r < n_ranks; r++) write_all(helper[r], id, sizeof id);
        }
    }

    if (argc >= 2 && !strcmp(argv[1], "--fake-jack")) {
        ASSERT(argc == 6);
        uint32_t nblocks = (uint32_t)strtoul(argv[2], NULL, 0), nframes = (uint32_t)strtoul(argv[3], NULL, 0);
        FILE *f = fopen(argv[4], "rb");
        ASSERT(f);
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        fake.nev = (size_t)sz / sizeof(struct fake_event);
        fake.ev = malloc(sz ? (size_t)sz : 1);
        ASSERT(fread(fake.ev, sizeof(struct fake_event), fake.nev, f) == fake.nev);
        fclose(f);
        fake.audio = malloc(sizeof(float) * nframes);
        fake.on = 1;
        jack.port_get_buffer = fake_port_get_buffer;
        jack.midi_get_event_count = fake_midi_get_event_count;
        jack.midi_event_get = fake_midi_event_get;
        midi_in = FAKE_MIDI_PORT;
        audio_out = FAKE_AUDIO_PORT;
        setup_bank(voices, id);
        synth_init(&synth);
        FILE *out = fopen(argv[5], "wb");
        ASSERT(out);
        double t_sum = 0, t_max = 0;                  /* wall time of process() per block */
        /* SYNTH_FAKE_PERIOD_US: pace the callbacks like a sound card would (e.g. 1333 for
           64 frames at 48 kHz); unset = back to back */
        const char *per = getenv("SYNTH_FAKE_PERIOD_US");
        const long period_ns = per ? strtol(per, NULL, 0) * 1000L : 0;
        struct timespec next;
        clock_gettime(CLOCK_MONOTONIC, &next);
        for (fake.block = 0; fake.block < nblocks; fake.block++) {
            fake.ncur = 0;
            while (fake.cursor < fake.nev && fake.ev[fake.cursor].block == fake.block && fake.ncur < 1024) {
                struct fake_event *e = &fake.ev[fake.cursor++];
                fake.cur[fake.ncur].time = 0;
                fake.cur[fake.ncur].size = e->size;
                fake.cur[fake.ncur].buffer = e->bytes;
                fake.ncur++;
            }
            /* more than 1024 events in one block: the surplus is dropped (a JACK MIDI port buffer is
               bounded too), never carried into a later block and never left to stall the cursor */
            while (fake.cursor < fake.nev && fake.ev[fake.cursor].block <= fake.block) fake.cursor++;
            struct timespec ta, tb;
            if (period_ns) {
                next.tv_nsec += period_ns;
                while (next.tv_nsec >= 1000000000L) { next.tv_nsec -= 1000000000L; next.tv_sec++; }
                clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &next, NULL);
            }
            clock_gettime(CLOCK_MONOTONIC, &ta);
            ASSERT(0 == process(nframes, NULL));
            clock_gettime(CLOCK_MONOTONIC, &tb);
            double us = (tb.tv_sec - ta.tv_sec) * 1e6 + (tb.tv_nsec - ta.tv_nsec) * 1e-3;
            if (fake.block >= 2) { t_sum += us; if (us > t_max) t_max = us; }   /* skip warm-up blocks */
            ASSERT(fwrite(fake.audio, sizeof(float), nframes, out) == nframes);
        }
        fclose(out);
        LOG("synth: fake-jack rendered %u blocks of %u frames, %u voices; process(): mean %.1f us, max %.1f us "
            "(deadline at 48 kHz: %.0f us)\n", nblocks, nframes, voices,
            nblocks > 2 ? t_sum / (nblocks - 2) : 0.0, t_max, nframes / 48000.0 * 1e6);
    } else {
        /* Jack client setup: linux/synth.c:285-301 */
        bind_jack();
        const char *client_name = "synth";
        int status = 0;
        jack_client_t *client = jack.client_open(client_name, JackNullOption, &status);
        ASSERT(client);
        ASSERT(midi_in = jack.port_register(client, "midi_in", JACK_DEFAULT_MIDI_TYPE, JackPortIsInput, 0));
        ASSERT(audio_out = jack.port_register(client, "audio_out", JACK_DEFAULT_AUDIO_TYPE, JackPortIsOutput, 0));
        setup_bank(voices, id);
        synth_init(&synth);
        jack.set_process_callback(client, process, 0);
        ASSERT(!mlockall(MCL_CURRENT | MCL_FUTURE));
        ASSERT(!jack.activate(client));
    }

    /* Input loop: only used to signal exit (linux/synth.c:304-310). */
    for (;;) {
        uint8_t buf[4];
        read_fixed(0, buf, sizeof(buf));
        exit(1);
    }
    return 0;
}
