/* The multi-GPU entry points of the C ABI (include/mi355x_sd.h, mi355x_sd_comm_*) from plain C, as ONE rank of a world of `world`
 * ranks: a compiled host's weight distribution (rank 0's packed buffer -> every rank, in place) and the final all-gather of the ranks'
 * latents. On a one-GPU box the test suite runs it as a world of one (tests/test_gpu_cexec.py): RCCL loaded through dlopen, a
 * communicator, both collectives on a stream, data checked. With several GPUs: start `world` processes, rank 0 writes the 128-byte id
 * to the file, the others read it.
 *
 *   gcc -std=c11 -O2 -I/opt/rocm/include -Iinclude tests/c/comm_test.c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,/opt/rocm/lib -o /tmp/comm_test
 *   LD_LIBRARY_PATH=paddlemix_amd /tmp/comm_test <rank> <world> <id-file> */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mi355x_sd.h"

#define CK(x)                                                                                         \
  do {                                                                                                \
    int rc_ = (x);                                                                                    \
    if (rc_) {                                                                                        \
      fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, mi355x_sd_last_error()); \
      return 2;                                                                                       \
    }                                                                                                 \
  } while (0)
#define HK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return 3;                                                                            \
    }                                                                                      \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s rank world id-file\n", argv[0]);
    return 64;
  }
  const int rank = atoi(argv[1]), world = atoi(argv[2]);
  CK(mi355x_sd_init(rank));   /* one GPU per rank */
  unsigned char id[128];
  if (rank == 0) {
    CK(mi355x_sd_comm_unique_id(id));
    FILE* f = fopen(argv[3], "wb");
    if (!f || fwrite(id, 1, 128, f) != 128) return 4;
    fclose(f);
  } else {
    FILE* f = NULL;
    for (int tries = 0; tries < 600 && !f; ++tries) {   /* wait for rank 0's file */
      f = fopen(argv[3], "rb");
      if (!f) {
        struct timespec ts = {0, 100000000};
        nanosleep(&ts, NULL);
      }
    }
    if (!f || fread(id, 1, 128, f) != 128) return 4;
    fclose(f);
  }
  void* comm = NULL;
  CK(mi355x_sd_comm_init(id, rank, world, &comm));
  hipStream_t st;
  HK(hipStreamCreate(&st));
  /* "weights": 8 MiB + 5 bytes (not a multiple of anything), rank 0 holds the pattern, everyone ends up with it */
  const size_t nw = (8u << 20) + 5;
  unsigned char *hw = malloc(nw), *dw = NULL;
  for (size_t i = 0; i < nw; ++i) hw[i] = rank == 0 ? (unsigned char)(i * 2654435761u >> 13) : 0xEE;
  HK(hipMalloc((void**)&dw, nw));
  HK(hipMemcpy(dw, hw, nw, hipMemcpyHostToDevice));
  CK(mi355x_sd_comm_broadcast(comm, dw, nw, 0, st));
  /* "latents": 4096 floats per rank, value = rank + i / 4096 */
  const size_t nl = 4096;
  float *hl = malloc(nl * 4), *hall = malloc(nl * 4 * world), *dl = NULL, *dall = NULL;
  for (size_t i = 0; i < nl; ++i) hl[i] = (float)rank + (float)i / (float)nl;
  HK(hipMalloc((void**)&dl, nl * 4));
  HK(hipMalloc((void**)&dall, nl * 4 * world));
  HK(hipMemcpy(dl, hl, nl * 4, hipMemcpyHostToDevice));
  CK(mi355x_sd_comm_all_gather(comm, dl, dall, nl * 4, st));
  HK(hipStreamSynchronize(st));
  HK(hipMemcpy(hw, dw, nw, hipMemcpyDeviceToHost));
  HK(hipMemcpy(hall, dall, nl * 4 * world, hipMemcpyDeviceToHost));
  int bad = 0;
  for (size_t i = 0; i < nw; ++i) bad += hw[i] != (unsigned char)(i * 2654435761u >> 13);
  for (int r = 0; r < world; ++r)
    for (size_t i = 0; i < nl; ++i) bad += hall[r * nl + i] != (float)r + (float)i / (float)nl;
  CK(mi355x_sd_comm_destroy(comm));
  printf("{\"rank\": %d, \"world\": %d, \"broadcast_bytes\": %zu, \"gathered_floats\": %zu, \"mismatches\": %d}\n", rank, world, nw, nl * world, bad);
  return bad ? 1 : 0;
}
