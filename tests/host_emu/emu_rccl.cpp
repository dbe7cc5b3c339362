// TEST HARNESS ONLY: the seven RCCL entry points csrc/od_comm.inc uses, over files in /dev/shm, so that the multi-process CPU tests run
// the product's od_comm_* code (argument checks, byte counts, block order) without a GPU.  Synchronous; the stream is ignored.
// An all-gather of sequence number s: every rank writes its block to <prefix>_<s>_<rank> (temporary name, then rename: readers never
// see a partial file), then reads the blocks of all ranks in rank order.  A rank can start gather s + 2 only after every rank has
// written its block of s + 1, i.e. has finished reading s: each rank removes its own block of s at the start of s + 2.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm { std::string prefix; int world, rank; long seq; };
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef void* hipStream_t_;

static const char* kErr[] = {"success", "emulated RCCL: system error", "emulated RCCL: timed out waiting for a peer", "emulated RCCL: invalid argument"};
const char* ncclGetErrorString(ncclResult_t r) { return kErr[r >= 0 && r < 4 ? r : 1]; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id->internal, 0, sizeof id->internal);
  FILE* f = std::fopen("/dev/urandom", "rb");
  unsigned char b[12];
  if (!f || std::fread(b, 1, sizeof b, f) != sizeof b) { if (f) std::fclose(f); return 1; }
  std::fclose(f);
  char* p = id->internal;
  p += std::sprintf(p, "odemu_%d_", (int)getpid());
  for (unsigned char c : b) p += std::sprintf(p, "%02x", c);
  return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* c, int world, ncclUniqueId id, int rank) {
  if (!c || world < 1 || rank < 0 || rank >= world || std::strncmp(id.internal, "odemu_", 6) != 0) return 3;
  *c = new ncclComm{std::string("/dev/shm/") + std::string(id.internal, strnlen(id.internal, 127)), world, rank, 0};
  return 0;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->world; return 0; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return 0; }

static std::string block_name(const ncclComm* c, long seq, int rank) { return c->prefix + "_" + std::to_string(seq) + "_" + std::to_string(rank); }

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t c, void* /*stream*/) {
  if (dt != 0) return 3;                                  // bytes only (ncclInt8): all od_comm.inc asks for
  const long s = c->seq++;
  if (s >= 2) std::remove(block_name(c, s - 2, c->rank).c_str());
  const std::string mine = block_name(c, s, c->rank), tmp = mine + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f || std::fwrite(send, 1, count, f) != count) { if (f) std::fclose(f); return 1; }
  std::fclose(f);
  if (std::rename(tmp.c_str(), mine.c_str()) != 0) return 1;
  for (int r = 0; r < c->world; ++r) {
    char* dst = (char*)recv + (size_t)r * count;
    if (r == c->rank) { std::memcpy(dst, send, count); continue; }
    const std::string nm = block_name(c, s, r);
    FILE* g = nullptr;
    for (int tries = 0; tries < 60000 && !(g = std::fopen(nm.c_str(), "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(2));
    if (!g) return 2;
    const size_t got = std::fread(dst, 1, count, g);
    std::fclose(g);
    if (got != count) return 1;
  }
  return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return 0;
  // (peers may still be reading the last two blocks: they are a few hundred bytes in the tests and are left to the test's cleanup of
  // /dev/shm/odemu_<pid>_*; blocks older than that were removed on the way)
  delete c;
  return 0;
}
}
