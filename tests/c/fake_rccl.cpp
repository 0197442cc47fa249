// fake_rccl.cpp -- TEST DOUBLE for librccl, never part of the product.
//
// The one-GPU test box cannot run the product's multi-rank path: RCCL refuses two ranks on one device.  This
// library is LD_PRELOADed in front of librccl by tests/test_multi_rank_gpu.py so that 2..4 PROCESSES sharing the
// one GPU can drive libsynth_mi355x.so's sharded bank exactly as they would on 2..4 GPUs: it implements the seven
// entry points the product calls (ncclGetUniqueId, ncclCommInitRank, ncclCommCount, ncclAllReduce, ncclAllGather,
// ncclGroupStart/End, ncclCommDestroy, ncclGetErrorString) over a POSIX shared-memory segment.
//   * ncclAllReduce(int32, sum) is stream-ordered like the real one: it waits for the stream, stages the operand
//     through shared memory, meets the other ranks at a barrier, sums, and writes the result back in place.
//   * It CHECKS what a real communicator would silently assume: every rank issues the same sequence of
//     collectives with the same counts (the SPMD contract of include/synth_mi355x.h); a mismatch aborts the rank
//     with a message, which fails the test.
// What it cannot show: latency, overlap, xGMI.  It shows that the product's queueing, grouping, contiguous-range
// arithmetic and pipelined hand-over produce the sum over ranks.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

namespace {
constexpr int MAX_RANKS = 8;
constexpr size_t MAX_COUNT = 1u << 20;          // int32 per rank per collective

struct Shm {
    std::atomic<int> attached;
    std::atomic<int> barrier_count;
    std::atomic<int> barrier_sense;
    std::atomic<unsigned long long> seq[MAX_RANKS];     // collectives issued by each rank
    unsigned long long count[MAX_RANKS];
    int32_t buf[MAX_RANKS][MAX_COUNT];
};

struct Comm {
    Shm *shm;
    int rank, nranks;
    int sense;
    char name[64];
};

[[noreturn]] void die(const char *what)
{
    fprintf(stderr, "fake_rccl: %s\n", what);
    fflush(stderr);
    _exit(97);
}

double now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

void barrier(Comm *c)
{
    Shm *s = c->shm;
    c->sense ^= 1;
    if (s->barrier_count.fetch_add(1) + 1 == c->nranks) {
        s->barrier_count.store(0);
        s->barrier_sense.store(c->sense);
    } else {
        const double t0 = now();
        while (s->barrier_sense.load() != c->sense) {
            sched_yield();
            if (now() - t0 > 60.0) die("a rank did not arrive at a collective within 60 s (SPMD contract broken?)");
        }
    }
}
}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/smx_fake_rccl_%d_%ld", (int)getpid(), (long)(now() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > MAX_RANKS) die("nranks out of range");
    Comm *c = new Comm();
    c->rank = rank; c->nranks = nranks; c->sense = 0;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) die("shm_open/ftruncate failed");
    c->shm = static_cast<Shm *>(mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    close(fd);
    if (c->shm == MAP_FAILED) die("mmap failed");
    c->shm->attached.fetch_add(1);
    const double t0 = now();
    while (c->shm->attached.load() < nranks) {            // a fresh segment is zero-filled
        sched_yield();
        if (now() - t0 > 60.0) die("not every rank called ncclCommInitRank within 60 s");
    }
    barrier(c);
    if (rank == 0) shm_unlink(c->name);                    // everybody has it mapped
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    *count = reinterpret_cast<Comm *>(comm)->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    Shm *s = c->shm;
    const bool is_max = datatype == ncclUint32 && op == ncclMax;            // sum_tick_square's OR of sign-bit words
    if (!(datatype == ncclInt32 && op == ncclSum) && !is_max) die("only int32 sums and uint32 maxima are used by the product");
    if (count == 0 || count > MAX_COUNT) die("count out of range");
    if (hipStreamSynchronize(stream) != hipSuccess) die("hipStreamSynchronize failed");
    if (hipMemcpy(s->buf[c->rank], sendbuff, count * 4, hipMemcpyDeviceToHost) != hipSuccess) die("D2H failed");
    const unsigned long long tagged = count | (is_max ? 1ull << 41 : 0ull);
    s->count[c->rank] = tagged;
    const unsigned long long my_seq = s->seq[c->rank].fetch_add(1) + 1;
    barrier(c);
    for (int r = 0; r < c->nranks; r++) {
        if (s->count[r] != tagged) {
            fprintf(stderr, "fake_rccl: rank %d issued a collective of %zu elements, rank %d one of %llu (collective #%llu)\n",
                    c->rank, count, r, s->count[r], my_seq);
            die("ranks issue different collectives: the SPMD contract is broken");
        }
        if (s->seq[r].load() != my_seq) die("ranks are at different collectives");
    }
    int32_t *sum = static_cast<int32_t *>(malloc(count * 4));
    for (size_t i = 0; i < count; i++) {
        uint32_t acc = 0;
        for (int r = 0; r < c->nranks; r++) {
            const uint32_t x = (uint32_t)s->buf[r][i];
            acc = is_max ? (x > acc ? x : acc) : acc + x;                        // the sum wraps, like the bus
        }
        sum[i] = (int32_t)acc;
    }
    barrier(c);                                             // everybody has read every operand
    if (hipMemcpy(recvbuff, sum, count * 4, hipMemcpyHostToDevice) != hipSuccess) die("H2D failed");
    free(sum);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    Shm *s = c->shm;
    if (datatype != ncclUint8) die("only byte all-gathers are used by the product");
    const size_t words = (sendcount + 3) / 4;
    if (sendcount == 0 || words > MAX_COUNT) die("all-gather size out of range");
    if (hipStreamSynchronize(stream) != hipSuccess) die("hipStreamSynchronize failed");
    if (hipMemcpy(s->buf[c->rank], sendbuff, sendcount, hipMemcpyDeviceToHost) != hipSuccess) die("D2H failed");
    s->count[c->rank] = sendcount | (1ull << 40);          // tagged: an all-gather, not an all-reduce
    const unsigned long long my_seq = s->seq[c->rank].fetch_add(1) + 1;
    barrier(c);
    for (int r = 0; r < c->nranks; r++) {
        if (s->count[r] != (sendcount | (1ull << 40))) die("ranks issue different collectives: the SPMD contract is broken");
        if (s->seq[r].load() != my_seq) die("ranks are at different collectives");
    }
    for (int r = 0; r < c->nranks; r++)
        if (hipMemcpy(static_cast<char *>(recvbuff) + (size_t)r * sendcount, s->buf[r], sendcount, hipMemcpyHostToDevice) != hipSuccess)
            die("H2D failed");
    barrier(c);
    return ncclSuccess;
}

}  // extern "C"
