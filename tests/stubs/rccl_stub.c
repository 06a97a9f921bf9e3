/* tests/stubs/rccl_stub.c -- TEST INFRASTRUCTURE.  A stand-in for librccl.so that exports the five entry points
 * epropnp/sharding.py:RcclComm binds (ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString)
 * with RCCL's signatures (rccl.h: ncclUniqueId is 128 bytes passed BY VALUE), implemented over a POSIX shared-memory segment
 * between the processes of one host.  It lets the direct-RCCL route of ObjectExchange -- the branch an 8-GPU node takes -- run
 * with 2 and 8 ranks on CPU tensors in the build container (tests/test_distributed.py), which a one-GPU box cannot do.
 *   RCCL_STUB_FAIL_RANK=<r>   ncclCommInitRank fails on that rank at once (the others give up after RCCL_STUB_TIMEOUT_MS, default
 *                             3000, as a bootstrap that lost a peer does) -- the partial-failure agreement of RcclComm
 * Built by the test: gcc -O1 -shared -fPIC rccl_stub.c -o librccl_stub.so -lrt */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 };

#define SLOT_BYTES (1u << 20)
#define MAX_RANKS 16
typedef struct {
  volatile int joined, left, arrive, generation;
  char pad[48];
  unsigned char data[MAX_RANKS][SLOT_BYTES];
} Segment;
typedef struct { Segment* seg; int nranks, rank; char name[64]; } Comm;

static long now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1000L + t.tv_nsec / 1000000L; }
static long timeout_ms(void) { const char* e = getenv("RCCL_STUB_TIMEOUT_MS"); return e ? atol(e) : 3000; }

const char* ncclGetErrorString(ncclResult_t r) {
  return r == ncclSuccess ? "no error" : (r == ncclSystemError ? "unhandled system error (stub)" : "invalid argument (stub)");
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static int counter = 0;
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/epropnp_rccl_stub_%d_%d_%ld", (int)getpid(), counter++, now_ms());
  return ncclSuccess;
}

static int barrier(Comm* c) {         /* sense-reversing; 0 = everyone arrived, -1 = timed out */
  Segment* s = c->seg;
  const int gen = __atomic_load_n(&s->generation, __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&s->arrive, 1, __ATOMIC_ACQ_REL) == c->nranks) {
    __atomic_store_n(&s->arrive, 0, __ATOMIC_RELEASE);
    __atomic_add_fetch(&s->generation, 1, __ATOMIC_ACQ_REL);
    return 0;
  }
  const long t0 = now_ms();
  while (__atomic_load_n(&s->generation, __ATOMIC_ACQUIRE) == gen) {
    if (now_ms() - t0 > 4 * timeout_ms()) return -1;
    usleep(50);
  }
  return 0;
}

ncclResult_t ncclCommInitRank(Comm** out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
  const char* fail = getenv("RCCL_STUB_FAIL_RANK");
  if (fail && atoi(fail) == rank) return ncclSystemError;
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) return ncclSystemError;
  Segment* seg = (Segment*)mmap(NULL, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (seg == MAP_FAILED) return ncclSystemError;
  __atomic_add_fetch(&seg->joined, 1, __ATOMIC_ACQ_REL);
  const long t0 = now_ms();
  while (__atomic_load_n(&seg->joined, __ATOMIC_ACQUIRE) < nranks) {       /* collective: every rank of the group must call */
    if (now_ms() - t0 > timeout_ms()) {
      munmap(seg, sizeof(Segment));
      if (rank == 0) shm_unlink(id.internal);
      return ncclSystemError;
    }
    usleep(50);
  }
  Comm* c = (Comm*)calloc(1, sizeof(Comm));
  c->seg = seg; c->nranks = nranks; c->rank = rank;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  *out = c;
  return ncclSuccess;
}

static size_t dtype_bytes(int dtype) { return dtype == 7 ? 4 : (dtype == 8 ? 8 : 0); }      /* ncclFloat32 = 7, ncclFloat64 = 8 */

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, int dtype, Comm* c, void* stream) {
  (void)stream;                     /* host memory: "in stream order" is program order */
  const size_t n = count * dtype_bytes(dtype);
  if (!c || !c->seg || n == 0 || n > SLOT_BYTES) return ncclInvalidArgument;
  memcpy((void*)c->seg->data[c->rank], send, n);
  if (barrier(c)) return ncclSystemError;
  for (int r = 0; r < c->nranks; ++r) memcpy((char*)recv + (size_t)r * n, (const void*)c->seg->data[r], n);
  if (barrier(c)) return ncclSystemError;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(Comm* c) {
  if (!c) return ncclInvalidArgument;
  if (c->seg) {
    const int left = __atomic_add_fetch(&c->seg->left, 1, __ATOMIC_ACQ_REL);
    munmap(c->seg, sizeof(Segment));
    if (left == c->nranks) shm_unlink(c->name);
  }
  free(c);
  return ncclSuccess;
}
