/* test_multi_rank.c -- a plain-C host that shards a saw bank over N processes (one per GPU; the one-GPU test box
 * puts them all on device 0 with tests/c/fake_rccl.cpp LD_PRELOADed, see tests/test_multi_rank_gpu.py), the way
 * INTEGRATION.md section 2 describes it for a C host: fork BEFORE any GPU call, rank 0 makes the 128-byte id and
 * hands it to the others over pipes (the host's own channel: no torch, no MPI), every rank runs its shard, the
 * library sums the buses (RCCL), and every rank gets the same samples.  ASSERT-based, exit status 0 on success
 * (the harness pattern of the reference's rules.mk:382-386).
 *
 * The check needs no oracle: the bank is linear in its voices, so the sum over ranks of N-way sharded banks must
 * equal what ONE process computes for the whole bank (rank 0 does that as well, without a communicator). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>
#include "synth_mi355x.h"

#define LOG(...) fprintf(stderr, __VA_ARGS__)
#define ASSERT(x) do { if (!(x)) { LOG("%s:%d: rank %d: ASSERT(%s) failed: %s\n", __FILE__, __LINE__, rank, #x, smx_last_error()); exit(1); } } while (0)

enum { PER = 6000, FRAMES = 64, BLOCKS = 12 };
static int rank = -1;

static void fill(uint32_t *inc, uint32_t *st, uint32_t first, uint32_t n) {
    for (uint32_t k = 0; k < n; k++) {
        uint32_t v = first + k, h = v * 2654435761u;
        inc[k] = (h % 10u == 0) ? 0u : note_to_inc(21 + (int)((h >> 12) % 88u));   /* a tenth of the voices off */
        st[k] = h * 40503u + 12345u;
    }
}

static int run_rank(int nranks, int device, const uint8_t *id, float *out /* BLOCKS*FRAMES */) {
    uint32_t *inc = malloc(PER * 4), *st = malloc(PER * 4);
    fill(inc, st, (uint32_t)rank * PER, PER);
    smx_bank *b = smx_bank_create(PER, device);
    ASSERT(b);
    ASSERT(0 == smx_bank_load(b, inc, st));
    ASSERT(0 == smx_bank_comm_init(b, rank, nranks, id));
    ASSERT(smx_bank_comm_ranks(b) == nranks);
    for (int blk = 0; blk < BLOCKS; blk++) {
        if (blk == BLOCKS / 2) ASSERT(0 == smx_bank_set_block_mode(b, SMX_BLOCK_PIPELINED));   /* every rank: collective */
        ASSERT(0 == smx_bank_run(b, out + blk * FRAMES, NULL, FRAMES));                        /* the SUM over ranks */
    }
    ASSERT(0 == smx_bank_sync(b));
    smx_bank_destroy(b);
    free(inc); free(st);
    return 0;
}

int main(int argc, char **argv) {
    const int nranks = argc > 1 ? atoi(argv[1]) : 2;
    const int same_device = argc > 2 ? atoi(argv[2]) : 0;       /* 1: every rank on device 0 (test double) */
    if (nranks < 1 || nranks > 8) { LOG("usage: %s NRANKS [SAME_DEVICE]\n", argv[0]); return 2; }
    int to_child[8][2], from_child[8][2];
    pid_t pid[8];
    /* fork first: no process may touch the GPU before it */
    for (int r = 1; r < nranks; r++) {
        if (pipe(to_child[r]) || pipe(from_child[r])) return 2;
        pid[r] = fork();
        if (pid[r] < 0) return 2;
        if (pid[r] == 0) {
            rank = r;
            uint8_t id[SMX_UNIQUE_ID_BYTES];
            ASSERT(read(to_child[r][0], id, sizeof id) == (ssize_t)sizeof id);
            float *out = malloc(sizeof(float) * BLOCKS * FRAMES);
            run_rank(nranks, same_device ? 0 : r, id, out);
            ASSERT(write(from_child[r][1], out, sizeof(float) * BLOCKS * FRAMES) == (ssize_t)(sizeof(float) * BLOCKS * FRAMES));
            _exit(0);
        }
    }
    rank = 0;
    uint8_t id[SMX_UNIQUE_ID_BYTES];
    ASSERT(0 == smx_comm_unique_id(id));
    for (int r = 1; r < nranks; r++) ASSERT(write(to_child[r][1], id, sizeof id) == (ssize_t)sizeof id);
    float *out = malloc(sizeof(float) * BLOCKS * FRAMES), *peer = malloc(sizeof(float) * BLOCKS * FRAMES);
    run_rank(nranks, 0, id, out);
    for (int r = 1; r < nranks; r++) {                          /* every rank holds the same reduced samples */
        ASSERT(read(from_child[r][0], peer, sizeof(float) * BLOCKS * FRAMES) == (ssize_t)(sizeof(float) * BLOCKS * FRAMES));
        ASSERT(0 == memcmp(out, peer, sizeof(float) * BLOCKS * FRAMES));
        int status = 0;
        waitpid(pid[r], &status, 0);
        ASSERT(WIFEXITED(status) && WEXITSTATUS(status) == 0);
    }
    /* the whole bank in ONE process, no communicator: the same samples, block for block (the pipelined half is
       one block late and starts with silence) */
    uint32_t n = (uint32_t)nranks * PER;
    uint32_t *inc = malloc(n * 4), *st = malloc(n * 4);
    fill(inc, st, 0, n);
    smx_bank *whole = smx_bank_create(n, 0);
    ASSERT(whole && 0 == smx_bank_load(whole, inc, st));
    float ref[BLOCKS][FRAMES];
    for (int blk = 0; blk < BLOCKS; blk++) ASSERT(0 == smx_bank_run(whole, ref[blk], NULL, FRAMES));
    for (int blk = 0; blk < BLOCKS / 2; blk++) ASSERT(0 == memcmp(out + blk * FRAMES, ref[blk], sizeof ref[blk]));
    for (int i = 0; i < FRAMES; i++) ASSERT(out[(BLOCKS / 2) * FRAMES + i] == 0.0f);
    for (int blk = BLOCKS / 2 + 1; blk < BLOCKS; blk++) ASSERT(0 == memcmp(out + blk * FRAMES, ref[blk - 1], sizeof ref[blk]));
    smx_bank_destroy(whole);
    LOG("test_multi_rank.c: %d ranks ok\n", nranks);
    return 0;
}
