// Multi-GPU entry points of the C ABI (include/mi355x_sd.h, "mi355x_sd_comm_*"): what a plain-C host needs for north_star's
// "RCCL broadcast of text-encoder / UNet weights over xGMI" -- one process per GPU, rank 0 holds the packed weight buffer
// (mi355x_sd_unet_finalize_weights / mi355x_sd_program_bind) and every other rank receives it in place; one all-gather of the ranks'
// latents at the end; nothing per step (prompts shard, batch rows never interact: SURVEY.md 8e). The reference's precedent does the
// same in-pipeline (PPD/pipelines/stable_diffusion_3/pipeline_stable_diffusion_3.py:803-839, its batch-parallel SD3 mode).
//
// RCCL is loaded with dlopen at the first comm call: the library has no link-time dependency on it and loads on machines that do
// not have it (every other entry point works there; these return MI355X_SD_ERR_UNSUPPORTED with a message). Host C++ only.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/mi355x_sd.h"
#include "kernels.h"

namespace {

// the subset of rccl.h this file uses (declared here so that the build needs no RCCL headers either)
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
struct Api {
  void* so = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
};
constexpr int kNcclInt8 = 0;   // ncclInt8 / ncclChar: the collectives here move bytes

int fail(int code, const std::string& msg) {
  sd::set_last_error(msg.c_str());
  return code;
}

Api& api() {
  static Api a = [] {
    Api x;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      x.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (x.so) break;
    }
    if (!x.so) {
      x.why = std::string("RCCL not found (dlopen librccl.so.1: ") + (dlerror() ? dlerror() : "?") + ")";
      return x;
    }
    auto sym = [&](const char* s) { return dlsym(x.so, s); };
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(sym("ncclBroadcast"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    if (!x.GetUniqueId || !x.CommInitRank || !x.CommDestroy || !x.Broadcast || !x.AllGather) x.why = "librccl.so lacks an expected symbol";
    return x;
  }();
  return a;
}

int nccl_fail(const char* what, int rc) {
  Api& a = api();
  return fail(MI355X_SD_ERR_HIP, std::string(what) + ": RCCL error " + std::to_string(rc) + " (" +
                                     (a.GetErrorString ? a.GetErrorString(rc) : "?") + ")");
}

struct CommH {
  Comm c = nullptr;
  int rank = 0, world = 0;
};

}  // namespace

extern "C" {

int mi355x_sd_comm_unique_id(void* id128) {
  if (!id128) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_comm_unique_id: null pointer");
  Api& a = api();
  if (!a.why.empty()) return fail(MI355X_SD_ERR_UNSUPPORTED, "mi355x_sd_comm_unique_id: " + a.why);
  UniqueId id;
  const int rc = a.GetUniqueId(&id);
  if (rc) return nccl_fail("mi355x_sd_comm_unique_id", rc);
  memcpy(id128, &id, sizeof(id));
  return MI355X_SD_OK;
}

int mi355x_sd_comm_init(const void* id128, int rank, int world, void** comm) {
  if (!id128 || !comm || world <= 0 || rank < 0 || rank >= world) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_comm_init: bad argument");
  Api& a = api();
  if (!a.why.empty()) return fail(MI355X_SD_ERR_UNSUPPORTED, "mi355x_sd_comm_init: " + a.why);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(MI355X_SD_ERR_HIP, "mi355x_sd_comm_init: no HIP device visible (select this rank's GPU with mi355x_sd_init first)");
  }
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  auto h = new CommH;
  h->rank = rank;
  h->world = world;
  const int rc = a.CommInitRank(&h->c, world, id, rank);
  if (rc) {
    delete h;
    return nccl_fail("mi355x_sd_comm_init", rc);
  }
  *comm = h;
  return MI355X_SD_OK;
}

int mi355x_sd_comm_broadcast(void* comm, void* buf, size_t bytes, int root, void* stream) {
  auto h = static_cast<CommH*>(comm);
  if (!h || !buf || root < 0 || root >= h->world) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_comm_broadcast: bad argument");
  if (!bytes) return MI355X_SD_OK;
  const int rc = api().Broadcast(buf, buf, bytes, kNcclInt8, root, h->c, static_cast<hipStream_t>(stream));
  return rc ? nccl_fail("mi355x_sd_comm_broadcast", rc) : MI355X_SD_OK;
}

int mi355x_sd_comm_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  auto h = static_cast<CommH*>(comm);
  if (!h || !send || !recv) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_comm_all_gather: bad argument");
  if (!bytes_per_rank) return MI355X_SD_OK;
  const int rc = api().AllGather(send, recv, bytes_per_rank, kNcclInt8, h->c, static_cast<hipStream_t>(stream));
  return rc ? nccl_fail("mi355x_sd_comm_all_gather", rc) : MI355X_SD_OK;
}

int mi355x_sd_comm_destroy(void* comm) {
  auto h = static_cast<CommH*>(comm);
  if (!h) return MI355X_SD_OK;
  const int rc = api().CommDestroy(h->c);
  delete h;
  return rc ? nccl_fail("mi355x_sd_comm_destroy", rc) : MI355X_SD_OK;
}

}  // extern "C"
