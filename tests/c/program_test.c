/* Plain-C client of seam B2 (include/mi355x_sd.h, mi355x_sd_program_*): no torch, no Python -- hipMalloc, the program handle, files.
 *
 *   gcc -std=c11 -O2 -I/opt/rocm/include -Iinclude tests/c/program_test.c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,/opt/rocm/lib -o /tmp/program_test
 *   LD_LIBRARY_PATH=paddlemix_amd /tmp/program_test <model.mi3prg> <inputs.bin> <outputs.bin> [use_graph] [runs]
 *
 * inputs.bin: the contents of every INPUT region in mi355x_sd_program_io_info order, back to back, each in the dtype the region
 * names (tests/test_gpu_export.py writes it); outputs.bin: every OUTPUT region in the same order. The comparison with the
 * Python-planned model that exported the program (bit equality) is the Python side's job. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mi355x_sd.h"

#define CK(x)                                                                                   \
  do {                                                                                          \
    int rc_ = (x);                                                                              \
    if (rc_) {                                                                                  \
      fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, mi355x_sd_last_error()); \
      return 2;                                                                                 \
    }                                                                                           \
  } while (0)
#define HK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return 3;                                                            \
    }                                                                      \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s model.mi3prg inputs.bin outputs.bin [use_graph] [runs]\n", argv[0]);
    return 1;
  }
  const int use_graph = argc > 4 ? atoi(argv[4]) : 0, runs = argc > 5 ? atoi(argv[5]) : 1;
  void* prog = NULL;
  CK(mi355x_sd_init(0));
  CK(mi355x_sd_program_load(argv[1], &prog));
  size_t bytes = 0;
  CK(mi355x_sd_program_device_bytes(prog, &bytes));
  void* dev = NULL;
  hipStream_t stream;
  HK(hipMalloc(&dev, bytes));
  HK(hipStreamCreate(&stream));
  CK(mi355x_sd_program_set_option(prog, "use_graph", use_graph));
  CK(mi355x_sd_program_bind(prog, dev, bytes, stream));
  const int n_io = mi355x_sd_program_num_io(prog);
  FILE* fin = fopen(argv[2], "rb");
  if (!fin) { perror(argv[2]); return 1; }
  for (int i = 0; i < n_io; ++i) {
    const char* name; int is_out, dtype, ndim; int64_t shape[4]; size_t nb; void* ptr;
    CK(mi355x_sd_program_io_info(prog, i, &name, &is_out, &dtype, shape, &ndim, &nb, &ptr));
    if (is_out) continue;
    void* host = malloc(nb);
    if (fread(host, 1, nb, fin) != nb) { fprintf(stderr, "inputs.bin too short at %s\n", name); return 1; }
    HK(hipMemcpyAsync(ptr, host, nb, hipMemcpyHostToDevice, stream));
    HK(hipStreamSynchronize(stream));
    free(host);
  }
  fclose(fin);
  for (int r = 0; r < runs; ++r) CK(mi355x_sd_program_run(prog, stream));
  HK(hipStreamSynchronize(stream));
  FILE* fout = fopen(argv[3], "wb");
  if (!fout) { perror(argv[3]); return 1; }
  for (int i = 0; i < n_io; ++i) {
    const char* name; int is_out, dtype, ndim; int64_t shape[4]; size_t nb; void* ptr;
    CK(mi355x_sd_program_io_info(prog, i, &name, &is_out, &dtype, shape, &ndim, &nb, &ptr));
    if (!is_out) continue;
    void* host = malloc(nb);
    HK(hipMemcpy(host, ptr, nb, hipMemcpyDeviceToHost));
    fwrite(host, 1, nb, fout);
    free(host);
    printf("output %s: %zu bytes\n", name, nb);
  }
  fclose(fout);
  printf("program_test: %d launches, %zu device bytes, %d run(s), use_graph=%d\n", mi355x_sd_program_num_launches(prog), bytes, runs, use_graph);
  CK(mi355x_sd_program_destroy(prog));
  HK(hipFree(dev));
  return 0;
}
