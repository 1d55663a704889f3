// Seam B2: an exported step program behind one handle (include/mi355x_sd.h, "mi355x_sd_program_*").
//
// The Python planners (paddlemix_amd/{unet,sd3,dit,vae,clip,t5}.py) turn a model + geometry into a static list of C-ABI launches
// over weights and scratch; paddlemix_amd/export.py writes that list to a file -- symbol names, arguments with every device
// pointer rewritten as (region, offset), the packed weights and plan-time constants -- and this runtime replays it for a host
// that has no Python: load (host only) -> device_bytes -> bind (caller's device buffer; uploads weights and constants) -> copy
// inputs to the named I/O regions -> run (eager or one hipGraph) -> read the outputs. The reference's own precedent for "export
// offline, run behind a predictor with named inputs": PPD/models/paddleinfer_runtime.py:47-126 + deploy/*/export_model.py.
//
// Host C++ only (the kernels are the per-op entry points this file calls); the library allocates no device memory.
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi355x_sd.h"
#include "kernels.h"

namespace {

enum ArgTag : uint32_t { A_INT = 0, A_FLOAT = 1, A_PTR = 2, A_NULL = 3, A_STREAM = 4 };
enum RegionKind : uint32_t { R_WEIGHT = 0, R_CONST = 1, R_SCRATCH = 2, R_IO = 3 };

struct PArg {
  uint32_t tag;
  int64_t i;     // A_INT value | A_PTR region index
  double f;      // A_FLOAT value
  uint64_t off;  // A_PTR byte offset
  void* p;       // resolved at bind (A_PTR) / run (A_STREAM)
};

// ---- typed marshaling: one PArg per C parameter, converted by the parameter's own type --------------------------------------
template <class T> struct Conv;
template <> struct Conv<int> { static constexpr char kind = 'i'; static int get(const PArg& a) { return (int)a.i; } };
template <> struct Conv<long> { static constexpr char kind = 'i'; static long get(const PArg& a) { return (long)a.i; } };
template <> struct Conv<unsigned long> { static constexpr char kind = 'i'; static unsigned long get(const PArg& a) { return (unsigned long)a.i; } };
template <> struct Conv<float> { static constexpr char kind = 'f'; static float get(const PArg& a) { return (float)a.f; } };
template <class T> struct Conv<T*> { static constexpr char kind = 'p'; static T* get(const PArg& a) { return (T*)a.p; } };

struct Entry {
  const char* name;
  std::string kinds;   // one of i / f / p per parameter
  std::function<int(const PArg*)> call;
};

template <class... A, size_t... I>
int invoke(int (*f)(A...), const PArg* a, std::index_sequence<I...>) { return f(Conv<A>::get(a[I])...); }

template <class... A>
Entry make_entry(const char* name, int (*f)(A...)) {
  return Entry{name, std::string{Conv<A>::kind...}, [f](const PArg* a) { return invoke(f, a, std::index_sequence_for<A...>{}); }};
}

#define SD_OP(fn) make_entry(#fn, &fn)
const std::vector<Entry>& op_table() {   // every stream-ordered launch of the header (the planners emit a subset)
  static const std::vector<Entry> t = {
      SD_OP(mi355x_sd_mask_to_bias), SD_OP(mi355x_sd_linear), SD_OP(mi355x_sd_linear_ex), SD_OP(mi355x_sd_row_stats),
      SD_OP(mi355x_sd_linear_ln), SD_OP(mi355x_sd_linear_f8), SD_OP(mi355x_sd_adaln_f8), SD_OP(mi355x_sd_linear_f8_q),
      SD_OP(mi355x_sd_quantize_rows), SD_OP(mi355x_sd_adaln), SD_OP(mi355x_sd_adaln_ex), SD_OP(mi355x_sd_patchify),
      SD_OP(mi355x_sd_unpatchify), SD_OP(mi355x_sd_conv3x3), SD_OP(mi355x_sd_sdpa), SD_OP(mi355x_sd_sdpa_ex),
      SD_OP(mi355x_sd_sdpa_accum), SD_OP(mi355x_sd_groupnorm_stats), SD_OP(mi355x_sd_groupnorm_act),
      SD_OP(mi355x_sd_scale_shift_act), SD_OP(mi355x_sd_layernorm), SD_OP(mi355x_sd_groupnorm_stats_ex),
      SD_OP(mi355x_sd_scale_shift_act_ex), SD_OP(mi355x_sd_layernorm_ex), SD_OP(mi355x_sd_cast_rows),
      SD_OP(mi355x_sd_fused_adaln_scale_residual), SD_OP(mi355x_sd_fused_adaln_scale_residual_ex), SD_OP(mi355x_sd_split_concat),
      SD_OP(mi355x_sd_timestep_embedding), SD_OP(mi355x_sd_silu), SD_OP(mi355x_sd_conv_in3x3), SD_OP(mi355x_sd_conv_in3x3_ex),
      SD_OP(mi355x_sd_conv_out3x3), SD_OP(mi355x_sd_copy_rows), SD_OP(mi355x_sd_add_nchw), SD_OP(mi355x_sd_add_nchw_ex),
      SD_OP(mi355x_sd_latent_dist), SD_OP(mi355x_sd_embed_tokens), SD_OP(mi355x_sd_activation), SD_OP(mi355x_sd_rmsnorm),
      SD_OP(mi355x_sd_gated_activation), SD_OP(mi355x_sd_conv1x1_nchw), SD_OP(mi355x_sd_softmax_rows), SD_OP(mi355x_sd_axpby),
      SD_OP(mi355x_sd_cfg_axpby)};
  return t;
}

struct Region {
  uint32_t kind;
  uint64_t bytes, data_off;   // data_off: position of the initial contents in the file (0: none)
  std::string name;
  uint64_t dev_off;           // assigned at load: offset inside the caller's device buffer
};
struct IO {
  uint32_t region, is_output, dtype, ndim;
  int64_t shape[4];
  std::string name;
};
struct Op {
  const Entry* e;
  std::vector<PArg> args;
};
struct Program {
  std::string path;
  uint32_t abi, elem;
  uint64_t workspace_bytes, device_bytes, workspace_off;
  std::vector<Region> regions;
  std::vector<IO> ios;
  std::vector<Op> ops;
  char* base = nullptr;       // bound device buffer
  void* graph = nullptr;      // hipGraphExec_t of the replay (use_graph)
  void* graph_stream = nullptr;
  int use_graph = 0;
};

int fail(int code, const std::string& msg) {
  sd::set_last_error(msg.c_str());
  return code;
}

struct Reader {
  FILE* f;
  bool ok = true;
  template <class T> T get() {
    T v{};
    if (ok && fread(&v, sizeof(T), 1, f) != 1) ok = false;
    return v;
  }
  std::string str() {
    uint32_t n = get<uint32_t>();
    if (!ok || n > 4096) { ok = false; return {}; }
    std::string s(n, '\0');
    if (n && fread(&s[0], 1, n, f) != n) ok = false;
    return s;
  }
};

constexpr uint64_t ALIGN = 256;
uint64_t round_up(uint64_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

}  // namespace

extern "C" {

int mi355x_sd_program_load(const char* path, void** handle) {
  if (!path || !handle) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_load: null pointer");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(MI355X_SD_ERR_INVALID, std::string("mi355x_sd_program_load: cannot open ") + path);
  Reader r{f};
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "MI3SDPRG", 8) != 0) {
    fclose(f);
    return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_load: not a program file (bad magic)");
  }
  auto p = new Program;
  p->path = path;
  const uint32_t version = r.get<uint32_t>();
  p->abi = r.get<uint32_t>();
  p->elem = r.get<uint32_t>();
  const uint32_t n_regions = r.get<uint32_t>(), n_ops = r.get<uint32_t>(), n_io = r.get<uint32_t>();
  p->workspace_bytes = r.get<uint64_t>();
  auto bail = [&](int code, const std::string& msg) {
    fclose(f);
    delete p;
    return fail(code, "mi355x_sd_program_load: " + msg);
  };
  if (!r.ok || version != 1) return bail(MI355X_SD_ERR_INVALID, "unsupported file version");
  if (p->abi != MI355X_SD_ABI_VERSION)
    return bail(MI355X_SD_ERR_UNSUPPORTED, "exported for ABI " + std::to_string(p->abi) + ", this library is ABI " +
                                              std::to_string(MI355X_SD_ABI_VERSION));
  if ((int)p->elem != mi355x_sd_elem_dtype())
    return bail(MI355X_SD_ERR_UNSUPPORTED, "exported for the other 16-bit element type (bf16 / fp16 builds do not mix)");
  if (n_regions > (1u << 20) || n_ops > (1u << 22) || n_io > 4096) return bail(MI355X_SD_ERR_INVALID, "implausible table sizes");
  uint64_t off = 0;
  for (uint32_t k = 0; k < n_regions && r.ok; ++k) {
    Region g;
    g.kind = r.get<uint32_t>();
    g.bytes = r.get<uint64_t>();
    g.data_off = r.get<uint64_t>();
    g.name = r.str();
    if (g.kind > R_IO) return bail(MI355X_SD_ERR_INVALID, "bad region kind");
    if (g.bytes > (1ull << 40)) return bail(MI355X_SD_ERR_INVALID, "implausible region size");
    g.dev_off = off;
    off += round_up(g.bytes);
    p->regions.push_back(std::move(g));
  }
  if (!r.ok || p->regions.size() != n_regions) return bail(MI355X_SD_ERR_INVALID, "truncated file (region table)");
  p->workspace_off = off;
  p->device_bytes = off + round_up(p->workspace_bytes);
  for (uint32_t k = 0; k < n_io && r.ok; ++k) {
    IO io;
    io.region = r.get<uint32_t>();
    io.is_output = r.get<uint32_t>();
    io.dtype = r.get<uint32_t>();
    io.ndim = r.get<uint32_t>();
    for (auto& s : io.shape) s = r.get<int64_t>();
    io.name = r.str();
    if (io.region >= n_regions || io.ndim > 4) return bail(MI355X_SD_ERR_INVALID, "bad I/O entry");
    p->ios.push_back(std::move(io));
  }
  if (!r.ok || p->ios.size() != n_io) return bail(MI355X_SD_ERR_INVALID, "truncated file (I/O table)");
  const auto& table = op_table();
  for (uint32_t k = 0; k < n_ops && r.ok; ++k) {
    const std::string name = r.str();
    const uint32_t nargs = r.get<uint32_t>();
    const Entry* e = nullptr;
    for (const auto& t : table)
      if (name == t.name) e = &t;
    if (!e) return bail(MI355X_SD_ERR_UNSUPPORTED, "op " + std::to_string(k) + ": unknown entry point '" + name + "'");
    if (nargs != e->kinds.size())
      return bail(MI355X_SD_ERR_INVALID, "op " + std::to_string(k) + " (" + name + "): " + std::to_string(nargs) + " arguments, the entry point takes " +
                                             std::to_string(e->kinds.size()));
    Op op{e, {}};
    for (uint32_t j = 0; j < nargs && r.ok; ++j) {
      PArg a{};
      a.tag = r.get<uint32_t>();
      const uint64_t u = r.get<uint64_t>(), v = r.get<uint64_t>();
      const char kind = e->kinds[j];
      bool good = false;
      switch (a.tag) {
        case A_INT: a.i = (int64_t)u; good = kind == 'i'; break;
        case A_FLOAT: memcpy(&a.f, &u, 8); good = kind == 'f'; break;
        case A_PTR:
          a.i = (int64_t)u; a.off = v;
          good = kind == 'p' && u < n_regions && v <= p->regions[u].bytes;
          break;
        case A_NULL: case A_STREAM: good = kind == 'p'; break;
        default: break;
      }
      if (!good)
        return bail(MI355X_SD_ERR_INVALID, "op " + std::to_string(k) + " (" + name + "), argument " + std::to_string(j) +
                                               ": tag does not fit the parameter's type, or pointer outside its region");
      op.args.push_back(a);
    }
    p->ops.push_back(std::move(op));
  }
  if (!r.ok) return bail(MI355X_SD_ERR_INVALID, "truncated file");
  fclose(f);
  *handle = p;
  return MI355X_SD_OK;
}

int mi355x_sd_program_destroy(void* handle) {
  auto p = static_cast<Program*>(handle);
  if (p) {
    if (p->graph) (void)mi355x_sd_graph_destroy(p->graph);
    delete p;
  }
  return MI355X_SD_OK;
}

int mi355x_sd_program_set_option(void* handle, const char* key, int value) {
  auto p = static_cast<Program*>(handle);
  if (!p || !key) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_set_option: null pointer");
  if (!strcmp(key, "use_graph")) {
    p->use_graph = value != 0;
    return MI355X_SD_OK;
  }
  return fail(MI355X_SD_ERR_UNSUPPORTED, std::string("mi355x_sd_program_set_option: unknown option ") + key);
}

int mi355x_sd_program_num_launches(void* handle) { return handle ? (int)static_cast<Program*>(handle)->ops.size() : -1; }

int mi355x_sd_program_device_bytes(void* handle, size_t* bytes) {
  auto p = static_cast<Program*>(handle);
  if (!p || !bytes) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_device_bytes: null pointer");
  *bytes = p->device_bytes;
  return MI355X_SD_OK;
}

int mi355x_sd_program_bind(void* handle, void* device_buffer, size_t bytes, void* stream) {
  auto p = static_cast<Program*>(handle);
  if (!p || !device_buffer) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_bind: null pointer");
  if (bytes < p->device_bytes || (reinterpret_cast<uintptr_t>(device_buffer) & (ALIGN - 1)))
    return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_bind: the device buffer must hold mi355x_sd_program_device_bytes() bytes, 256-byte aligned");
  FILE* f = fopen(p->path.c_str(), "rb");
  if (!f) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_bind: cannot reopen " + p->path);
  std::vector<char> stage;
  char* base = static_cast<char*>(device_buffer);
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (const auto& g : p->regions) {
    if (!g.data_off || !g.bytes) continue;
    stage.resize(g.bytes);
    if (fseek(f, (long)g.data_off, SEEK_SET) != 0 || fread(stage.data(), 1, g.bytes, f) != g.bytes) {
      fclose(f);
      return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_bind: truncated data of region " + g.name);
    }
    if (hipMemcpyAsync(base + g.dev_off, stage.data(), g.bytes, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {   // the staging vector is reused: one region in flight at a time
      fclose(f);
      return fail(MI355X_SD_ERR_HIP, "mi355x_sd_program_bind: upload of region " + g.name + " failed");
    }
  }
  fclose(f);
  for (auto& op : p->ops)
    for (auto& a : op.args)
      if (a.tag == A_PTR) a.p = base + p->regions[(size_t)a.i].dev_off + a.off;
  if (p->graph) {
    (void)mi355x_sd_graph_destroy(p->graph);
    p->graph = nullptr;
  }
  p->base = base;
  return MI355X_SD_OK;
}

int mi355x_sd_program_num_io(void* handle) { return handle ? (int)static_cast<Program*>(handle)->ios.size() : -1; }

int mi355x_sd_program_io_info(void* handle, int index, const char** name, int* is_output, int* dtype, int64_t* shape4, int* ndim,
                              size_t* bytes, void** device_ptr) {
  auto p = static_cast<Program*>(handle);
  if (!p || index < 0 || index >= (int)p->ios.size()) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_io_info: bad handle / index");
  const IO& io = p->ios[(size_t)index];
  const Region& g = p->regions[io.region];
  if (name) *name = io.name.c_str();
  if (is_output) *is_output = (int)io.is_output;
  if (dtype) *dtype = (int)io.dtype;
  if (shape4) memcpy(shape4, io.shape, sizeof(io.shape));
  if (ndim) *ndim = (int)io.ndim;
  if (bytes) *bytes = g.bytes;
  if (device_ptr) *device_ptr = p->base ? p->base + g.dev_off : nullptr;   // NULL until bind
  return MI355X_SD_OK;
}

int mi355x_sd_program_run(void* handle, void* stream) {
  auto p = static_cast<Program*>(handle);
  if (!p) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_run: null handle");
  if (!p->base) return fail(MI355X_SD_ERR_INVALID, "mi355x_sd_program_run: mi355x_sd_program_bind first");
  // (the split-K / widening scratch the program was planned with is one of its scratch regions since ABI 12 -- an argument of every
  // GEMM-class launch, same size -> same split decisions -> same bits as the exporting process; nothing process-wide is bound here)
  int rc = MI355X_SD_OK;
  auto replay = [&]() {
    for (size_t k = 0; k < p->ops.size(); ++k) {
      auto& op = p->ops[k];
      for (auto& a : op.args)
        if (a.tag == A_STREAM) a.p = stream;
      int r = op.e->call(op.args.data());
      if (r) {
        std::string why = mi355x_sd_last_error();
        return fail(r, "mi355x_sd_program_run: launch " + std::to_string(k) + " (" + op.e->name + ") failed: " + why);
      }
    }
    return (int)MI355X_SD_OK;
  };
  if (!p->use_graph) return replay();
  if (p->graph && p->graph_stream != stream) {
    (void)mi355x_sd_graph_destroy(p->graph);
    p->graph = nullptr;
  }
  if (!p->graph) {
    rc = replay();   // once outside capture (lazy module loading), like the Python host
    if (rc) return rc;
    rc = mi355x_sd_graph_begin(stream);
    if (rc) return rc;
    const int rr = replay();
    void* exec = nullptr;
    rc = mi355x_sd_graph_end(stream, &exec);
    if (rr || rc) {   // (a replay that failed under capture: the instantiated graph, if any, is not kept)
      if (exec) (void)mi355x_sd_graph_destroy(exec);
      return rr ? rr : rc;
    }
    p->graph = exec;
    p->graph_stream = stream;
  }
  return mi355x_sd_graph_launch(p->graph, stream);
}

}  // extern "C"
