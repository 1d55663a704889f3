// Seam B1 in the C ABI: UNet2DConditionModel as an opaque handle -- config in, weights in, one call per denoising step.
//
//   mi355x_sd_unet_create(config_json, &h)            parse the reference's config.json, build the layer list
//   mi355x_sd_unet_load_weight(h, name, ptr, ...)     reference parameter names, Paddle layouts (Linear [in,out], conv OIHW), host memory
//   mi355x_sd_unet_weight_bytes / _finalize_weights   pack once (transposes, fused QKV, batched K/V and time projections, GEGLU
//                                                     interleave) into a CALLER-OWNED device buffer
//   mi355x_sd_unet_plan(h, B, H, W, L, &bytes)        static program + workspace layout for one input geometry
//   mi355x_sd_unet_bind_workspace(h, ptr, bytes)      caller-owned device arena; nothing is allocated on the device by the library
//   mi355x_sd_unet_forward(h, stream, ...)            stage inputs + replay the program (optionally as a hipGraph)
//
// It is the C++ form of paddlemix_amd/unet.py (same op sequence, same packing -> bit-identical results, tested) for the
// configurations of the hot path: the four SD block types, mid cross-attention block, conv or linear projections,
// addition_embed_type None | "text_time", 16-bit or fp32 residual stream. Reference: UNet2DConditionModel.forward
// (ppdiffusers/ppdiffusers/models/unet_2d_condition.py:809-1207); the opaque-predictor precedent for this seam is
// PaddleInferRuntimeModel.__call__ (ppdiffusers/ppdiffusers/models/paddleinfer_runtime.py:47-126) with the named inputs of
// ppdiffusers/deploy/sd15/export_model.py:78-90 (sample, timestep, encoder_hidden_states [+ text_embeds, time_ids]).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/mi355x_sd.h"
#include "common.h"
#include "kernels.h"

namespace {

using sd::bf16;

// ---------------------------------------------------------------------------------------------------------------- errors
struct ExecError {
  int code;
  std::string msg;
};
[[noreturn]] void die(int code, const std::string& m) { throw ExecError{code, m}; }

// ---------------------------------------------------------------------------------------------------------------- tiny JSON
struct JVal {
  enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal* get(const std::string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
};
struct JParser {
  const char* p;
  int depth = 0;   // nesting of the value being parsed: a model config nests two deep; a million '[' must not be a stack overflow
  explicit JParser(const char* s) : p(s) {}
  void ws() {
    while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') ++p;
  }
  JVal parse() {
    struct Depth {
      int& d;
      explicit Depth(int& d_) : d(d_) {
        if (++d > 32) die(MI355X_SD_ERR_INVALID, "config_json: nested more than 32 levels deep");
      }
      ~Depth() { --d; }
    } guard(depth);
    ws();
    JVal v;
    if (*p == '{') {
      v.kind = JVal::OBJ;
      ++p;
      ws();
      if (*p == '}') { ++p; return v; }
      for (;;) {
        ws();
        JVal k = parse();
        if (k.kind != JVal::STR) die(MI355X_SD_ERR_INVALID, "config_json: object key is not a string");
        ws();
        if (*p != ':') die(MI355X_SD_ERR_INVALID, "config_json: expected ':'");
        ++p;
        v.obj.emplace_back(k.str, parse());
        ws();
        if (*p == ',') { ++p; continue; }
        if (*p == '}') { ++p; return v; }
        die(MI355X_SD_ERR_INVALID, "config_json: expected ',' or '}'");
      }
    }
    if (*p == '[') {
      v.kind = JVal::ARR;
      ++p;
      ws();
      if (*p == ']') { ++p; return v; }
      for (;;) {
        v.arr.push_back(parse());
        ws();
        if (*p == ',') { ++p; continue; }
        if (*p == ']') { ++p; return v; }
        die(MI355X_SD_ERR_INVALID, "config_json: expected ',' or ']'");
      }
    }
    if (*p == '"') {
      v.kind = JVal::STR;
      ++p;
      while (*p && *p != '"') {
        if (*p == '\\' && p[1]) ++p;
        v.str.push_back(*p++);
      }
      if (*p != '"') die(MI355X_SD_ERR_INVALID, "config_json: unterminated string");
      ++p;
      return v;
    }
    if (!strncmp(p, "true", 4)) { p += 4; v.kind = JVal::BOOL; v.b = true; return v; }
    if (!strncmp(p, "false", 5)) { p += 5; v.kind = JVal::BOOL; v.b = false; return v; }
    if (!strncmp(p, "null", 4)) { p += 4; return v; }
    char* end = nullptr;
    v.num = strtod(p, &end);
    if (end == p) die(MI355X_SD_ERR_INVALID, "config_json: unexpected character");
    p = end;
    v.kind = JVal::NUM;
    return v;
  }
};

// ---------------------------------------------------------------------------------------------------------------- config
struct Cfg {
  int in_channels = 4, out_channels = 4;
  bool flip_sin_to_cos = true;
  double freq_shift = 0.0;
  std::vector<std::string> down = {"CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"};
  std::vector<std::string> up = {"UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"};
  std::vector<int> boc = {320, 640, 1280, 1280};
  std::vector<int> layers_per_block, tlayers, heads;
  int cross_dim = 1280;
  double mid_scale = 1.0;
  int groups = 32;
  double norm_eps = 1e-5;
  bool linear_proj = false;
  bool text_time = false;
  int atd = 0, pdim = 0;
  // class embedding (unet_2d_condition.py:404-441, 953-975) and TimestepEmbedding.cond_proj (embeddings.py:265-266, 284-285)
  enum ClassKind { CLASS_NONE, CLASS_TABLE, CLASS_TIMESTEP, CLASS_IDENTITY, CLASS_PROJECTION, CLASS_SIMPLE };
  ClassKind class_kind = CLASS_NONE;
  int num_class = 0;      // rows of the nn.Embedding table (CLASS_TABLE)
  int tcp = 0;            // time_cond_proj_dim (0: none)
  bool concat = false;    // class_embeddings_concat: the blocks see [emb | class_emb] (blocks_time_embed_dim = 2 x time_embed_dim, :443-449)
  // IP-Adapter (encoder_hid_dim_type "ip_image_proj": ImageProjection embeddings.py:507-518 + IPAdapterAttnProcessor on every attn2)
  bool ip = false;
  int ehd = 0, ip_tokens = 4;   // encoder_hid_dim; ImageProjection.num_image_text_embeds (extension key ip_adapter_num_tokens, as unet.py)
};

// A config number that has to be an integer inside [lo, hi]: the text comes from a host (a file someone edited), and a width of
// 2^31 or a depth of 10^30 must be a refusal here -- not an out-of-range float-to-int conversion, and not a parameter table that
// exhausts the machine's memory before anything else can object (tests/test_boundary_fuzz.py).
int as_int(double x, const char* key, int lo, int hi) {
  if (!(x >= (double)lo && x <= (double)hi) || x != floor(x))
    die(MI355X_SD_ERR_INVALID, std::string("config ") + key + ": expected an integer between " + std::to_string(lo) + " and " + std::to_string(hi));
  return (int)x;
}

std::vector<int> int_or_list(const JVal* v, size_t n, int dflt, const char* key, int lo, int hi) {
  std::vector<int> out(n, dflt);
  if (!v || v->kind == JVal::NUL) return out;
  if (v->kind == JVal::NUM) return std::vector<int>(n, as_int(v->num, key, lo, hi));
  if (v->kind != JVal::ARR || v->arr.size() != n) die(MI355X_SD_ERR_INVALID, std::string("config ") + key + ": expected an int or one int per block");
  for (size_t i = 0; i < n; ++i) {
    if (v->arr[i].kind != JVal::NUM) die(MI355X_SD_ERR_UNSUPPORTED, std::string("config ") + key + ": nested lists are not implemented");
    out[i] = as_int(v->arr[i].num, key, lo, hi);
  }
  return out;
}

Cfg parse_config(const char* json) {
  JParser jp(json);
  const JVal root = jp.parse();
  if (root.kind != JVal::OBJ) die(MI355X_SD_ERR_INVALID, "config_json: expected an object");
  Cfg c;
  auto num = [&](const char* k, double d) {
    const JVal* v = root.get(k);
    return (v && v->kind == JVal::NUM) ? v->num : d;
  };
  auto boolean = [&](const char* k, bool d) {
    const JVal* v = root.get(k);
    return (v && v->kind == JVal::BOOL) ? v->b : d;
  };
  auto strs = [&](const char* k, std::vector<std::string> d) {
    const JVal* v = root.get(k);
    if (!v || v->kind != JVal::ARR) return d;
    std::vector<std::string> o;
    for (auto& e : v->arr) o.push_back(e.str);
    return o;
  };
  // what this executor does not build must not load silently (same refusals as paddlemix_amd/unet.py normalize_config)
  static const char* must_be_null[] = {
                                       "time_embedding_dim", "time_embedding_act_fn", "timestep_post_act", "cross_attention_norm",
                                       "mid_block_only_cross_attention", "reverse_transformer_layers_per_block", "num_attention_heads"};
  for (const char* k : must_be_null) {
    const JVal* v = root.get(k);
    if (v && !(v->kind == JVal::NUL || (v->kind == JVal::BOOL && !v->b)))
      die(MI355X_SD_ERR_UNSUPPORTED, std::string("mi355x_sd_unet_create: config ") + k + " is not implemented by the C executor");
  }
  static const char* must_be_false[] = {"center_input_sample", "dual_cross_attention", "only_cross_attention", "resnet_skip_time_act"};
  for (const char* k : must_be_false)
    if (boolean(k, false)) die(MI355X_SD_ERR_UNSUPPORTED, std::string("mi355x_sd_unet_create: config ") + k + "=true is not implemented");
  auto str_is = [&](const char* k, const char* want) {
    const JVal* v = root.get(k);
    if (v && v->kind == JVal::STR && v->str != want)
      die(MI355X_SD_ERR_UNSUPPORTED, std::string("mi355x_sd_unet_create: config ") + k + "=" + v->str + " is not implemented");
  };
  str_is("act_fn", "silu");
  str_is("time_embedding_type", "positional");
  str_is("resnet_time_scale_shift", "default");
  str_is("attention_type", "default");
  str_is("mid_block_type", "UNetMidBlock2DCrossAttn");
  str_is("data_format", "NCHW");
  if (const JVal* v = root.get("mid_block_type"))   // null = "no mid block" in the reference (unet_2d_condition.py:553-555)
    if (v->kind == JVal::NUL) die(MI355X_SD_ERR_UNSUPPORTED, "mi355x_sd_unet_create: config mid_block_type=null (no mid block) is not implemented");
  if (boolean("upcast_attention", false)) die(MI355X_SD_ERR_UNSUPPORTED, "mi355x_sd_unet_create: config upcast_attention=true is not implemented");
  if (num("downsample_padding", 1) != 1 || num("conv_in_kernel", 3) != 3 || num("conv_out_kernel", 3) != 3 ||
      num("resnet_out_scale_factor", 1.0) != 1.0 || num("dropout", 0.0) != 0.0)
    die(MI355X_SD_ERR_UNSUPPORTED, "mi355x_sd_unet_create: downsample_padding / conv kernels / resnet_out_scale_factor / dropout off default");

  c.in_channels = as_int(num("in_channels", 4), "in_channels", 1, 64);
  c.out_channels = as_int(num("out_channels", 4), "out_channels", 1, 64);
  c.flip_sin_to_cos = boolean("flip_sin_to_cos", true);
  c.freq_shift = num("freq_shift", 0);
  c.down = strs("down_block_types", c.down);
  c.up = strs("up_block_types", c.up);
  if (const JVal* v = root.get("block_out_channels")) {
    if (v->kind != JVal::ARR) die(MI355X_SD_ERR_INVALID, "config block_out_channels: expected a list");
    c.boc.clear();
    for (auto& e : v->arr) {
      if (e.kind != JVal::NUM) die(MI355X_SD_ERR_INVALID, "config block_out_channels: expected integers");
      if (c.boc.size() >= 8) die(MI355X_SD_ERR_INVALID, "config block_out_channels: at most 8 levels");
      c.boc.push_back(as_int(e.num, "block_out_channels", 8, 16384));
    }
  }
  const size_t n = c.down.size();
  if (c.boc.size() != n || c.up.size() != n || n < 1) die(MI355X_SD_ERR_INVALID, "config: block lists must have one entry per level");
  for (auto& bt : c.down)
    if (bt != "CrossAttnDownBlock2D" && bt != "DownBlock2D") die(MI355X_SD_ERR_UNSUPPORTED, "block type " + bt + " is not implemented");
  for (auto& bt : c.up)
    if (bt != "CrossAttnUpBlock2D" && bt != "UpBlock2D") die(MI355X_SD_ERR_UNSUPPORTED, "block type " + bt + " is not implemented");
  c.layers_per_block = int_or_list(root.get("layers_per_block"), n, 2, "layers_per_block", 1, 16);
  c.tlayers = int_or_list(root.get("transformer_layers_per_block"), n, 1, "transformer_layers_per_block", 1, 64);
  c.heads = int_or_list(root.get("attention_head_dim"), n, 8, "attention_head_dim", 1, 2048);   // the naming quirk at unet_2d_condition.py:245
  std::vector<int> cross = int_or_list(root.get("cross_attention_dim"), n, 1280, "cross_attention_dim", 8, 65536);
  for (int x : cross)
    if (x != cross[0]) die(MI355X_SD_ERR_UNSUPPORTED, "per-block cross_attention_dim is not implemented");
  c.cross_dim = cross[0];
  c.mid_scale = num("mid_block_scale_factor", 1.0);
  c.groups = as_int(num("norm_num_groups", 32), "norm_num_groups", 1, 1024);
  c.norm_eps = num("norm_eps", 1e-5);
  c.linear_proj = boolean("use_linear_projection", false);
  if (const JVal* v = root.get("addition_embed_type")) {
    if (v->kind == JVal::STR) {
      if (v->str != "text_time") die(MI355X_SD_ERR_UNSUPPORTED, "addition_embed_type=" + v->str + " is not implemented");
      c.text_time = true;
      c.atd = as_int(num("addition_time_embed_dim", 0), "addition_time_embed_dim", 0, 8192);
      c.pdim = as_int(num("projection_class_embeddings_input_dim", 0), "projection_class_embeddings_input_dim", 0, 1 << 20);
      if (c.atd <= 0 || c.pdim <= 0) die(MI355X_SD_ERR_INVALID, "text_time needs addition_time_embed_dim and projection_class_embeddings_input_dim");
    }
  }
  // projection_class_embeddings_input_dim also sizes the "projection" / "simple_projection" class embeddings
  if (const JVal* v = root.get("projection_class_embeddings_input_dim"))
    if (v->kind == JVal::NUM) c.pdim = as_int(v->num, "projection_class_embeddings_input_dim", 1, 1 << 20);
  if (const JVal* v = root.get("time_cond_proj_dim"))
    if (v->kind != JVal::NUL) {
      if (v->kind != JVal::NUM) die(MI355X_SD_ERR_INVALID, "config time_cond_proj_dim: expected an integer or null");
      c.tcp = as_int(v->num, "time_cond_proj_dim", 8, 1 << 16);
      if (c.tcp & 7) die(MI355X_SD_ERR_UNSUPPORTED, "time_cond_proj_dim must be a multiple of 8");
    }
  {
    const JVal* ct = root.get("class_embed_type");
    const JVal* nc = root.get("num_class_embeds");
    const bool has_nc = nc && nc->kind != JVal::NUL;
    if (ct && ct->kind != JVal::NUL) {
      if (ct->kind != JVal::STR) die(MI355X_SD_ERR_INVALID, "config class_embed_type: expected a string or null");
      if (ct->str == "timestep") c.class_kind = Cfg::CLASS_TIMESTEP;
      else if (ct->str == "identity") c.class_kind = Cfg::CLASS_IDENTITY;
      else if (ct->str == "projection") c.class_kind = Cfg::CLASS_PROJECTION;
      else if (ct->str == "simple_projection") c.class_kind = Cfg::CLASS_SIMPLE;
      else die(MI355X_SD_ERR_INVALID, "class_embed_type '" + ct->str + "'");   // the reference's ValueError (unet_2d_condition.py:439-441)
      if ((c.class_kind == Cfg::CLASS_PROJECTION || c.class_kind == Cfg::CLASS_SIMPLE) && c.pdim <= 0)
        die(MI355X_SD_ERR_INVALID, "`class_embed_type`: '" + ct->str + "' requires `projection_class_embeddings_input_dim` be set");
      if ((c.class_kind == Cfg::CLASS_PROJECTION || c.class_kind == Cfg::CLASS_SIMPLE) && (c.pdim & 7))
        die(MI355X_SD_ERR_UNSUPPORTED, "projection_class_embeddings_input_dim must be a multiple of 8");
    } else if (has_nc) {
      if (nc->kind != JVal::NUM) die(MI355X_SD_ERR_INVALID, "config num_class_embeds: expected an integer or null");
      c.class_kind = Cfg::CLASS_TABLE;
      c.num_class = as_int(nc->num, "num_class_embeds", 1, 1 << 24);
    }
    c.concat = boolean("class_embeddings_concat", false);
    if (c.concat && c.class_kind == Cfg::CLASS_NONE)   // the reference builds 2x-wide time_emb_proj and then feeds them the 1x embedding
      die(MI355X_SD_ERR_INVALID, "class_embeddings_concat=true needs a class embedding (class_embed_type / num_class_embeds)");
    if (c.concat && c.text_time) die(MI355X_SD_ERR_UNSUPPORTED, "class_embeddings_concat together with addition_embed_type is not implemented");
  }
  {
    const JVal* ty = root.get("encoder_hid_dim_type");
    const JVal* hd = root.get("encoder_hid_dim");
    const bool has_hd = hd && hd->kind != JVal::NUL;
    if (ty && ty->kind != JVal::NUL) {
      if (ty->kind != JVal::STR || ty->str != "ip_image_proj")   // text_proj / text_image_proj / image_proj (unet_2d_condition.py:300-335) are not built
        die(MI355X_SD_ERR_UNSUPPORTED, "mi355x_sd_unet_create: config encoder_hid_dim_type=" + (ty->kind == JVal::STR ? ty->str : std::string("?")) +
                                           " is not implemented");
      if (!has_hd) die(MI355X_SD_ERR_INVALID, "`encoder_hid_dim` has to be defined when `encoder_hid_dim_type` is set to ip_image_proj.");
      if (hd->kind != JVal::NUM) die(MI355X_SD_ERR_INVALID, "config encoder_hid_dim: expected an integer");
      c.ip = true;
      c.ehd = as_int(hd->num, "encoder_hid_dim", 8, 1 << 16);
      if (c.ehd & 7) die(MI355X_SD_ERR_UNSUPPORTED, "encoder_hid_dim must be a multiple of 8");
      c.ip_tokens = as_int(num("ip_adapter_num_tokens", 4), "ip_adapter_num_tokens", 1, 4096);
    } else if (has_hd) {
      die(MI355X_SD_ERR_UNSUPPORTED, "mi355x_sd_unet_create: config encoder_hid_dim without a type (text_proj) is not implemented");
    }
  }
  if (c.out_channels > 4) die(MI355X_SD_ERR_UNSUPPORTED, "out_channels > 4");
  if (c.in_channels <= 0 || c.out_channels <= 0 || c.cross_dim <= 0 || (c.cross_dim & 7))
    die(MI355X_SD_ERR_INVALID, "config: in_channels / out_channels / cross_attention_dim must be positive (cross_attention_dim a multiple of 8)");
  if (c.groups <= 0) die(MI355X_SD_ERR_INVALID, "config norm_num_groups must be positive");
  for (size_t i = 0; i < n; ++i) {
    // GroupNorm groups of whole channels, 16-byte channel chunks in every kernel, and heads that divide the width
    if (c.boc[i] <= 0 || c.boc[i] % c.groups || (c.boc[i] & 7))
      die(MI355X_SD_ERR_INVALID, "config block_out_channels: every entry must be a positive multiple of norm_num_groups and of 8");
    if (c.layers_per_block[i] <= 0 || c.tlayers[i] <= 0) die(MI355X_SD_ERR_INVALID, "config layers_per_block / transformer_layers_per_block must be positive");
    if (c.heads[i] <= 0 || c.boc[i] % c.heads[i] || ((c.boc[i] / c.heads[i]) & 7) || c.boc[i] / c.heads[i] > 160)
      die(MI355X_SD_ERR_INVALID, "config attention_head_dim: the head count must be positive and divide block_out_channels into head "
                                 "dims that are multiples of 8 and at most 160");
  }
  return c;
}

// ---------------------------------------------------------------------------------------------------------------- structure
struct Layer {
  enum Kind { SKIP, RESNET, ATTN, DOWN, UP, CAT } kind;
  std::string name;
  int cin = 0, cout = 0;   // RESNET: cin -> cout; ATTN / DOWN / UP: channels in cout; CAT: skip channels in cout
  double scale = 1.0;
  int heads = 0, layers = 0;
};

std::vector<Layer> structure(const Cfg& c) {   // == paddlemix_amd/unet.py:_structure
  std::vector<Layer> L;
  const int n = (int)c.boc.size();
  auto S = [](int v) { return std::to_string(v); };
  L.push_back({Layer::SKIP});
  int out_c = c.boc[0];
  for (int i = 0; i < n; ++i) {
    const int in_c = out_c;
    out_c = c.boc[i];
    for (int j = 0; j < c.layers_per_block[i]; ++j) {
      L.push_back({Layer::RESNET, "down_blocks." + S(i) + ".resnets." + S(j), j == 0 ? in_c : out_c, out_c, 1.0});
      if (c.down[i] == "CrossAttnDownBlock2D")
        L.push_back({Layer::ATTN, "down_blocks." + S(i) + ".attentions." + S(j), 0, out_c, 1.0, c.heads[i], c.tlayers[i]});
      L.push_back({Layer::SKIP});
    }
    if (i != n - 1) {
      L.push_back({Layer::DOWN, "down_blocks." + S(i) + ".downsamplers.0.conv", 0, out_c});
      L.push_back({Layer::SKIP});
    }
  }
  L.push_back({Layer::RESNET, "mid_block.resnets.0", c.boc[n - 1], c.boc[n - 1], c.mid_scale});
  L.push_back({Layer::ATTN, "mid_block.attentions.0", 0, c.boc[n - 1], 1.0, c.heads[n - 1], c.tlayers[n - 1]});
  L.push_back({Layer::RESNET, "mid_block.resnets.1", c.boc[n - 1], c.boc[n - 1], c.mid_scale});
  out_c = c.boc[n - 1];
  for (int i = 0; i < n; ++i) {
    const int ri = n - 1 - i;
    const int prev_out = out_c;
    out_c = c.boc[ri];
    const int in_c = c.boc[n - 1 - std::min(i + 1, n - 1)];
    const int nl = c.layers_per_block[ri] + 1;
    for (int j = 0; j < nl; ++j) {
      const int skip_c = (j == nl - 1) ? in_c : out_c;
      const int rin = (j == 0) ? prev_out : out_c;
      L.push_back({Layer::CAT, "", 0, skip_c});
      L.push_back({Layer::RESNET, "up_blocks." + S(i) + ".resnets." + S(j), rin + skip_c, out_c, 1.0});
      if (c.up[i] == "CrossAttnUpBlock2D")
        L.push_back({Layer::ATTN, "up_blocks." + S(i) + ".attentions." + S(j), 0, out_c, 1.0, c.heads[ri], c.tlayers[ri]});
    }
    if (i != n - 1) L.push_back({Layer::UP, "up_blocks." + S(i) + ".upsamplers.0.conv", 0, out_c});
  }
  return L;
}

// ---------------------------------------------------------------------------------------------------------------- weights
struct HostT {
  std::vector<int64_t> shape;
  std::vector<float> v;
  bool loaded = false;
  int64_t numel() const {
    int64_t n = 1;
    for (auto d : shape) n *= d;
    return n;
  }
};

float bf16_bits_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
float f16_bits_to_f32(uint16_t h) {
  _Float16 x;
  memcpy(&x, &h, 2);
  return (float)x;
}
// fp32 -> the build's 16-bit element, round to nearest even (what torch's .to(bfloat16 / float16) does)
uint16_t to_elem16(float f) {
#ifdef MI355X_SD_F16
  _Float16 x = (_Float16)f;
  uint16_t h;
  memcpy(&h, &x, 2);
  return h;
#else
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
#endif
}

struct Packed {   // one packed device tensor: offset into the weight buffer
  size_t off = 0, bytes = 0;
  int rows = 0, cols = 0;   // [N][K] for matrices
};

// symbolic workspace address
struct Ref {
  int buf = -1;   // index into Exec::bufs; -1: absolute (weights / null)
  size_t off = 0;
  const void* abs = nullptr;
};
struct View {   // rows x C elements of `es` bytes, row stride ld elements
  Ref p;
  int rows = 0, C = 0, ld = 0, es = 2;
  View cols(int off, int c) const {
    View v = *this;
    v.p.off += (size_t)off * es;
    v.C = c;
    return v;
  }
};

struct Exec {
  Cfg cfg;
  std::vector<Layer> layers;
  std::map<std::string, HostT> params;            // expected parameters (reference names, Paddle layouts)
  std::vector<std::string> order;                 // construction order
  bool resid_f32 = false;
  bool fold_scale = true;                         // head_dim^-0.5 * log2(e) folded into the self-attention to_q weights (head_dim 64)
  std::set<std::string> log2_blocks;
  // packed weights
  std::map<std::string, Packed> w;
  std::vector<unsigned char> host_pack;
  bool packed = false;        // Packer ran (it consumes the loaded fp32 tensors: it must never run twice)
  size_t weight_bytes = 0;
  unsigned char* dev_w = nullptr;
  std::map<std::string, int> temb_off, kv_off;
  std::set<std::string> kb64;                     // 3x3 convs whose weights are packed in 64-channel blocks
  int temb_total = 0, kv_total = 0;
  // plan
  int B = 0, H = 0, W = 0, L = 0;
  struct Buf {
    std::string name;
    size_t bytes = 0, off = 0;
  };
  std::vector<Buf> bufs;
  std::map<std::string, int> buf_index;
  size_t workspace_bytes = 0;
  unsigned char* ws = nullptr;
  std::vector<std::function<int(void*)>> prog_sym;   // built at plan time, read `ws` at run time
  Ref in_sample, in_t, in_scale, in_enc, in_addin, in_tids, out;
  Ref in_class, in_tcond, in_image;               // class_labels / timestep_cond / image_embeds as the program reads them (mi355x_sd_unet_set_input)
  const float* image_ptr = nullptr;
  float ip_scale = 1.0f;                          // IPAdapterAttnProcessor.scale (mi355x_sd_unet_set_ip_adapter_scale; a launch constant of the plan)
  const void* class_ptr = nullptr;                // caller's device tensors, read by every forward call until replaced
  const float* tcond_ptr = nullptr;
  // optional inputs chosen at plan time (mi355x_sd_unet_plan_ex flags)
  int plan_flags = 0;
  Ref enc_mask, enc_bias, self_mask, self_bias, ctrl_mid;
  std::vector<Ref> ctrl_down;
  struct SkipShape { int C, h, w; };
  std::vector<SkipShape> skip_shapes;             // every skip tensor, then the mid block's output
  int text_dim = 0, n_ids = 0;
  bool planned = false, consts_set = false;
  hipGraphExec_t graph = nullptr;
  void* graph_stream = nullptr;

  void* at(const Ref& r) const { return r.buf < 0 ? const_cast<void*>(r.abs) : (void*)(ws + bufs[r.buf].off + r.off); }
  void drop_plan() {
    planned = false;
    prog_sym.clear();
    bufs.clear();
    buf_index.clear();
    skip_shapes.clear();
    ctrl_down.clear();
    ws = nullptr;
  }
};

void expect(Exec& e, const std::string& name, std::vector<int64_t> shape) {
  e.params[name].shape = std::move(shape);
  e.order.push_back(name);
}

void build_param_table(Exec& e) {   // == unet_param_shapes for the supported configs
  const Cfg& c = e.cfg;
  const int ted = c.boc[0] * 4;
  auto lin = [&](const std::string& n, int i, int o, bool bias = true) {
    expect(e, n + ".weight", {i, o});
    if (bias) expect(e, n + ".bias", {o});
  };
  auto conv = [&](const std::string& n, int i, int o, int k) {
    expect(e, n + ".weight", {o, i, k, k});
    expect(e, n + ".bias", {o});
  };
  auto norm = [&](const std::string& n, int ch) {
    expect(e, n + ".weight", {ch});
    expect(e, n + ".bias", {ch});
  };
  conv("conv_in", c.in_channels, c.boc[0], 3);
  lin("time_embedding.linear_1", c.boc[0], ted);
  if (c.tcp) lin("time_embedding.cond_proj", c.tcp, c.boc[0], false);
  lin("time_embedding.linear_2", ted, ted);
  if (c.class_kind == Cfg::CLASS_TABLE) {
    expect(e, "class_embedding.weight", {c.num_class, ted});   // nn.Embedding [classes, dim]: a table, never transposed
  } else if (c.class_kind == Cfg::CLASS_TIMESTEP || c.class_kind == Cfg::CLASS_PROJECTION) {
    lin("class_embedding.linear_1", c.class_kind == Cfg::CLASS_TIMESTEP ? c.boc[0] : c.pdim, ted);
    lin("class_embedding.linear_2", ted, ted);
  } else if (c.class_kind == Cfg::CLASS_SIMPLE) {
    lin("class_embedding", c.pdim, ted);
  }
  if (c.text_time) {
    lin("add_embedding.linear_1", c.pdim, ted);
    lin("add_embedding.linear_2", ted, ted);
  }
  for (auto& d : e.layers) {
    if (d.kind == Layer::RESNET) {
      norm(d.name + ".norm1", d.cin);
      conv(d.name + ".conv1", d.cin, d.cout, 3);
      lin(d.name + ".time_emb_proj", ted * (c.concat ? 2 : 1), d.cout);
      norm(d.name + ".norm2", d.cout);
      conv(d.name + ".conv2", d.cout, d.cout, 3);
      if (d.cin != d.cout) conv(d.name + ".conv_shortcut", d.cin, d.cout, 1);
    } else if (d.kind == Layer::ATTN) {
      const int ch = d.cout;
      norm(d.name + ".norm", ch);
      if (c.linear_proj) lin(d.name + ".proj_in", ch, ch);
      else conv(d.name + ".proj_in", ch, ch, 1);
      for (int l = 0; l < d.layers; ++l) {
        const std::string b = d.name + ".transformer_blocks." + std::to_string(l);
        norm(b + ".norm1", ch);
        for (int a = 0; a < 2; ++a) {
          const std::string an = b + (a == 0 ? ".attn1" : ".attn2");
          const int kd = a == 0 ? ch : c.cross_dim;
          lin(an + ".to_q", ch, ch, false);
          lin(an + ".to_k", kd, ch, false);
          lin(an + ".to_v", kd, ch, false);
          lin(an + ".to_out.0", ch, ch);
          if (a == 0) {
            norm(b + ".norm2", ch);
          } else if (c.ip) {   // IPAdapterAttnProcessor (attention_processor.py:1816-1817)
            lin(an + ".processor.to_k_ip", kd, ch, false);
            lin(an + ".processor.to_v_ip", kd, ch, false);
          }
        }
        norm(b + ".norm3", ch);
        lin(b + ".ff.net.0.proj", ch, 8 * ch);
        lin(b + ".ff.net.2", 4 * ch, ch);
      }
      if (c.linear_proj) lin(d.name + ".proj_out", ch, ch);
      else conv(d.name + ".proj_out", ch, ch, 1);
    } else if (d.kind == Layer::DOWN || d.kind == Layer::UP) {
      conv(d.name, d.cout, d.cout, 3);
    }
  }
  norm("conv_norm_out", c.boc[0]);
  conv("conv_out", c.boc[0], c.out_channels, 3);
  if (c.ip) {   // ImageProjection; last, like unet_param_shapes
    lin("encoder_hid_proj.image_embeds", c.ehd, c.ip_tokens * c.cross_dim);
    norm("encoder_hid_proj.norm", c.cross_dim);
  }
}

// ---- packing (== UNet2DConditionModel._load_weights) ----
struct Packer {
  Exec& e;
  explicit Packer(Exec& ex) : e(ex) {}
  const HostT& get(const std::string& n) {
    auto it = e.params.find(n);
    if (it == e.params.end() || !it->second.loaded) die(MI355X_SD_ERR_INVALID, "missing parameter " + n);
    return it->second;
  }
  // every packed tensor is staged in its own buffer (pointers stay valid while others are added) and laid out at the end
  std::map<std::string, std::vector<unsigned char>> staged;
  std::vector<std::string> staged_order;
  unsigned char* reserve(const std::string& key, size_t bytes, int rows, int cols) {
    if (staged.count(key)) die(MI355X_SD_ERR_INVALID, "internal: packed tensor " + key + " defined twice");
    std::vector<unsigned char>& v = staged[key];
    v.resize(bytes);
    staged_order.push_back(key);
    e.w[key] = Packed{0, bytes, rows, cols};
    return v.data();
  }
  uint16_t* m16(const std::string& key, int rows, int cols) {
    return reinterpret_cast<uint16_t*>(reserve(key, (size_t)rows * cols * 2, rows, cols));
  }
  float* f32(const std::string& key, int n) { return reinterpret_cast<float*>(reserve(key, (size_t)n * 4, 1, n)); }
  // Paddle Linear [in, out] -> rows [out][in] appended at row r0 of dst (row length in); mul: fp32 factor applied before rounding
  static void lin_rows(const HostT& t, uint16_t* dst, int r0, float mul = 1.0f) {
    const int in = (int)t.shape[0], out = (int)t.shape[1];
    for (int o = 0; o < out; ++o)
      for (int i = 0; i < in; ++i) dst[(size_t)(r0 + o) * in + i] = to_elem16(mul == 1.0f ? t.v[(size_t)i * out + o] : t.v[(size_t)i * out + o] * mul);
  }
  void put_vec(const std::string& key, const std::string& name) {
    const HostT& t = get(name);
    memcpy(f32(key, (int)t.numel()), t.v.data(), t.numel() * 4);
  }
  void put_lin(const std::string& key, const std::string& name, bool bias = true) {
    const HostT& t = get(name + ".weight");
    lin_rows(t, m16(key + ".w", (int)t.shape[1], (int)t.shape[0]), 0);
    if (bias) put_vec(key + ".b", name + ".bias");
  }
  // OIHW -> [O][kh][kw][I]; 3x3 with I % 64 == 0 -> [O][I/64][kh][kw][64] (MI355X_SD_CONV_KB64: taps innermost per 64-channel block)
  void put_conv(const std::string& key, const std::string& name) {
    const HostT& t = get(name + ".weight");
    const int O = (int)t.shape[0], I = (int)t.shape[1], kh = (int)t.shape[2], kw = (int)t.shape[3];
    const bool kb64 = kh == 3 && (I % 64) == 0;
    if (kb64) e.kb64.insert(key);
    uint16_t* d = m16(key + ".w", O, kh * kw * I);
    for (int o = 0; o < O; ++o)
      for (int i = 0; i < I; ++i)
        for (int y = 0; y < kh; ++y)
          for (int x = 0; x < kw; ++x) {
            const size_t dst = kb64 ? ((((size_t)o * (I / 64) + i / 64) * kh + y) * kw + x) * 64 + i % 64
                                    : (((size_t)o * kh + y) * kw + x) * I + i;
            d[dst] = to_elem16(t.v[(((size_t)o * I + i) * kh + y) * kw + x]);
          }
    put_vec(key + ".b", name + ".bias");
  }
  void put_norm(const std::string& key, const std::string& name) {
    put_vec(key + ".g", name + ".weight");
    put_vec(key + ".b", name + ".bias");
  }
  void run() {
    const Cfg& c = e.cfg;
    {   // conv_in: -> [ky][kx][ci][O]
      const HostT& t = get("conv_in.weight");
      const int O = (int)t.shape[0], I = (int)t.shape[1];
      uint16_t* d = m16("conv_in.w", 9 * I, O);
      for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
          for (int y = 0; y < 3; ++y)
            for (int x = 0; x < 3; ++x) d[(((size_t)y * 3 + x) * I + i) * O + o] = to_elem16(t.v[(((size_t)o * I + i) * 3 + y) * 3 + x]);
      put_vec("conv_in.b", "conv_in.bias");
    }
    put_lin("time_embedding.linear_1", "time_embedding.linear_1");
    put_lin("time_embedding.linear_2", "time_embedding.linear_2");
    if (c.tcp) put_lin("time_embedding.cond_proj", "time_embedding.cond_proj", false);
    if (c.class_kind == Cfg::CLASS_TABLE) {
      const HostT& t = get("class_embedding.weight");
      uint16_t* d = m16("class_embedding.table", (int)t.shape[0], (int)t.shape[1]);
      for (size_t i = 0; i < t.numel(); ++i) d[i] = to_elem16(t.v[i]);
    } else if (c.class_kind == Cfg::CLASS_TIMESTEP || c.class_kind == Cfg::CLASS_PROJECTION) {
      put_lin("class_embedding.linear_1", "class_embedding.linear_1");
      put_lin("class_embedding.linear_2", "class_embedding.linear_2");
    } else if (c.class_kind == Cfg::CLASS_SIMPLE) {
      put_lin("class_embedding", "class_embedding");
    }
    if (c.text_time) {
      put_lin("add_embedding.linear_1", "add_embedding.linear_1");
      put_lin("add_embedding.linear_2", "add_embedding.linear_2");
    }
    // sizes of the batched matrices first
    const int ted = c.boc[0] * 4;
    int toff = 0, koff = 0;
    for (auto& d : e.layers) {
      if (d.kind == Layer::RESNET) {
        e.temb_off[d.name] = toff;
        toff += d.cout;
      } else if (d.kind == Layer::ATTN) {
        for (int l = 0; l < d.layers; ++l) {
          e.kv_off[d.name + ".transformer_blocks." + std::to_string(l)] = koff;
          koff += 2 * d.cout;
        }
      }
    }
    e.temb_total = toff;
    e.kv_total = koff;
    for (auto& d : e.layers) {
      if (d.kind == Layer::RESNET) {
        put_norm(d.name + ".norm1", d.name + ".norm1");
        put_conv(d.name + ".conv1", d.name + ".conv1");
        put_norm(d.name + ".norm2", d.name + ".norm2");
        put_conv(d.name + ".conv2", d.name + ".conv2");
        if (d.cin != d.cout) put_conv(d.name + ".conv_shortcut", d.name + ".conv_shortcut");
      } else if (d.kind == Layer::ATTN) {
        const int ch = d.cout;
        put_norm(d.name + ".norm", d.name + ".norm");
        if (c.linear_proj) {
          put_lin(d.name + ".proj_in", d.name + ".proj_in");
          put_lin(d.name + ".proj_out", d.name + ".proj_out");
        } else {
          put_conv(d.name + ".proj_in", d.name + ".proj_in");
          put_conv(d.name + ".proj_out", d.name + ".proj_out");
        }
        for (int l = 0; l < d.layers; ++l) {
          const std::string b = d.name + ".transformer_blocks." + std::to_string(l);
          for (const char* nm : {".norm1", ".norm2", ".norm3"}) put_norm(b + nm, b + nm);
          uint16_t* q = m16(b + ".attn1.qkv.w", 3 * ch, ch);
          float qmul = 1.0f;
          if (e.fold_scale && ch / d.heads == 64) {
            qmul = (float)(pow((double)(ch / d.heads), -0.5) * 1.4426950408889634);
            e.log2_blocks.insert(b);
          }
          lin_rows(get(b + ".attn1.to_q.weight"), q, 0, qmul);
          lin_rows(get(b + ".attn1.to_k.weight"), q, ch);
          lin_rows(get(b + ".attn1.to_v.weight"), q, 2 * ch);
          lin_rows(get(b + ".attn2.to_q.weight"), m16(b + ".attn2.q.w", ch, ch), 0);
          put_lin(b + ".attn1.out", b + ".attn1.to_out.0");
          put_lin(b + ".attn2.out", b + ".attn2.to_out.0");
          {   // GEGLU: rows interleaved [16 value | 16 gate]
            const HostT& t = get(b + ".ff.net.0.proj.weight");   // [c, 8c]
            const HostT& tb = get(b + ".ff.net.0.proj.bias");
            const int in = (int)t.shape[0], out = (int)t.shape[1], half = out / 2;
            uint16_t* d1 = m16(b + ".ff1.w", out, in);
            float* b1 = f32(b + ".ff1.b", out);
            for (int o = 0; o < out; ++o) {
              const int src = o < half ? o : o - half;            // value rows come from [0, half), gate rows from [half, out)
              const int grp = src / 16, r = src % 16;
              const int dst = grp * 32 + (o < half ? 0 : 16) + r;
              for (int i = 0; i < in; ++i) d1[(size_t)dst * in + i] = to_elem16(t.v[(size_t)i * out + o]);
              b1[dst] = tb.v[o];
            }
          }
          put_lin(b + ".ff2", b + ".ff.net.2");
        }
      } else if (d.kind == Layer::DOWN || d.kind == Layer::UP) {
        put_conv(d.name, d.name);
      }
    }
    {   // every cross-attention to_k / to_v in one matrix
      uint16_t* kv = m16("kv_all.w", e.kv_total, c.cross_dim);
      for (auto& d : e.layers)
        if (d.kind == Layer::ATTN)
          for (int l = 0; l < d.layers; ++l) {
            const std::string b = d.name + ".transformer_blocks." + std::to_string(l);
            const int r0 = e.kv_off[b];
            lin_rows(get(b + ".attn2.to_k.weight"), kv, r0);
            lin_rows(get(b + ".attn2.to_v.weight"), kv, r0 + d.cout);
          }
    }
    if (c.ip) {   // ... and every IPAdapterAttnProcessor's to_k_ip / to_v_ip at the same row offsets
      uint16_t* kv = m16("kvip_all.w", e.kv_total, c.cross_dim);
      for (auto& d : e.layers)
        if (d.kind == Layer::ATTN)
          for (int l = 0; l < d.layers; ++l) {
            const std::string b = d.name + ".transformer_blocks." + std::to_string(l);
            const int r0 = e.kv_off[b];
            lin_rows(get(b + ".attn2.processor.to_k_ip.weight"), kv, r0);
            lin_rows(get(b + ".attn2.processor.to_v_ip.weight"), kv, r0 + d.cout);
          }
      put_lin("encoder_hid_proj.image_embeds", "encoder_hid_proj.image_embeds");
      put_norm("encoder_hid_proj.norm", "encoder_hid_proj.norm");
    }
    {   // every resnet time_emb_proj in one matrix
      uint16_t* tw = m16("temb_all.w", e.temb_total, ted * (c.concat ? 2 : 1));
      float* tb = f32("temb_all.b", e.temb_total);
      for (auto& d : e.layers)
        if (d.kind == Layer::RESNET) {
          const int r0 = e.temb_off[d.name];
          lin_rows(get(d.name + ".time_emb_proj.weight"), tw, r0);
          const HostT& bb = get(d.name + ".time_emb_proj.bias");
          memcpy(tb + r0, bb.v.data(), bb.numel() * 4);
        }
    }
    put_norm("conv_norm_out", "conv_norm_out");
    put_conv("conv_out", "conv_out");
    size_t off = 0;
    for (auto& key : staged_order) {
      e.w[key].off = off;
      off += (staged[key].size() + 255) & ~(size_t)255;
    }
    e.weight_bytes = off;
    e.host_pack.assign(off, 0);
    for (auto& key : staged_order) {
      memcpy(e.host_pack.data() + e.w[key].off, staged[key].data(), staged[key].size());
      std::vector<unsigned char>().swap(staged[key]);
    }
    for (auto& kv : e.params) {   // the fp32 host copies are no longer needed
      std::vector<float>().swap(kv.second.v);
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------- plan
struct Planner {
  Exec& e;
  const Cfg& c;
  int B, H, W, L;
  const int RES;   // bytes per element of the residual stream
  explicit Planner(Exec& ex) : e(ex), c(ex.cfg), B(ex.B), H(ex.H), W(ex.W), L(ex.L), RES(ex.resid_f32 ? 4 : 2) {}

  Ref sc(const std::string& name, size_t bytes) {   // named scratch, sized to its maximum use
    auto it = e.buf_index.find(name);
    int idx;
    if (it == e.buf_index.end()) {
      idx = (int)e.bufs.size();
      e.bufs.push_back({name, 0, 0});
      e.buf_index[name] = idx;
    } else {
      idx = it->second;
    }
    if (bytes > e.bufs[idx].bytes) e.bufs[idx].bytes = bytes;
    Ref r;
    r.buf = idx;
    return r;
  }
  int persist_n = 0;
  Ref persist(size_t bytes) { return sc("persist." + std::to_string(persist_n++), bytes < 16 ? 16 : bytes); }
  // the split-K scratch of this handle's GEMM-class launches: part of the handle's own arena (ABI 12: an argument of every such
  // call, not a process-wide binding). Same size as the Python planner's (paddlemix_amd/program.py WORKSPACE_BYTES): the split
  // count depends on it, and the two planners are held bit-identical (tests/test_gpu_cexec.py)
  static constexpr size_t GEMM_WS_BYTES = (size_t)32 << 20;
  Ref gemm_ws() { return sc("gemm_ws", GEMM_WS_BYTES); }
  Ref wref(const std::string& key) {
    auto it = e.w.find(key);
    if (it == e.w.end()) die(MI355X_SD_ERR_INVALID, "internal: packed weight " + key + " missing");
    Ref r;
    r.abs = e.dev_w + it->second.off;
    return r;
  }
  const Packed& wp(const std::string& key) { return e.w.at(key); }
  View view(Ref p, int rows, int C, int es = 2) {
    View v;
    v.p = p;
    v.rows = rows;
    v.C = C;
    v.ld = C;
    v.es = es;
    return v;
  }
  void emit(std::function<int(void*)> f) { e.prog_sym.push_back(std::move(f)); }

  void linear(const View& a, const std::string& wkey, const View& out, bool bias = true, const View* R = nullptr, int flags = 0,
              float out_scale = 1.0f, Ref rowbias = Ref(), int rpb = 0, int ld_rb = 0) {
    const Packed& w = wp(wkey + ".w");
    const int N = w.rows, K = w.cols;
    if (K != a.C || a.es != 2) die(MI355X_SD_ERR_INVALID, "internal: linear " + wkey + " operand mismatch");
    flags |= (out.es == 4 ? MI355X_SD_OUT_F32 : 0) | ((R && R->es == 4) ? MI355X_SD_R_F32 : 0);
    Exec* ex = &e;
    const Ref wr = wref(wkey + ".w");
    Ref br;
    if (bias) br = wref(wkey + ".b");
    const bool has_rb = rowbias.buf >= 0 || rowbias.abs;
    const bool has_R = R != nullptr;
    const View Rv = R ? *R : View();
    const Ref gw = gemm_ws();
    emit([=](void* st) {
      return mi355x_sd_linear(ex->at(a.p), a.ld, ex->at(wr), ex->at(out.p), out.ld, a.rows, N, K, bias ? (const float*)ex->at(br) : nullptr,
                              has_rb ? (const float*)ex->at(rowbias) : nullptr, rpb, ld_rb, has_R ? ex->at(Rv.p) : nullptr, has_R ? Rv.ld : 0,
                              out_scale, flags, ex->at(gw), GEMM_WS_BYTES, st);
    });
  }
  void conv3(const View& x, int h, int w_, const std::string& wkey, const View& out, int stride = 1, int up = 0, Ref rowbias = Ref(),
             const View* R = nullptr, float out_scale = 1.0f) {
    const Packed& w = wp(wkey + ".w");
    const int Cout = w.rows;
    if (x.es != 2) die(MI355X_SD_ERR_INVALID, "internal: conv operand is fp32");
    const int flags = (out.es == 4 ? MI355X_SD_OUT_F32 : 0) | ((R && R->es == 4) ? MI355X_SD_R_F32 : 0) |
                      (e.kb64.count(wkey) ? MI355X_SD_CONV_KB64 : 0);
    Exec* ex = &e;
    const Ref wr = wref(wkey + ".w"), br = wref(wkey + ".b");
    const bool has_rb = rowbias.buf >= 0 || rowbias.abs;
    const bool has_R = R != nullptr;
    const View Rv = R ? *R : View();
    const int Bc = B, tt = e.temb_total;
    const Ref gw = gemm_ws();
    emit([=](void* st) {
      return mi355x_sd_conv3x3(ex->at(x.p), x.ld, Bc, h, w_, x.C, stride, up, ex->at(wr), ex->at(out.p), out.ld, Cout, (const float*)ex->at(br),
                               has_rb ? (const float*)ex->at(rowbias) : nullptr, has_rb ? tt : 0, has_R ? ex->at(Rv.p) : nullptr,
                               has_R ? Rv.ld : 0, out_scale, flags, ex->at(gw), GEMM_WS_BYTES, st);
    });
  }
  View gnorm(const View& x, int hw, const std::string& nkey, float eps, bool silu, const View* raw16 = nullptr) {
    const int nws = mi355x_sd_groupnorm_workspace_floats(B, hw, x.C);
    const Ref ws = sc("gn_ws", (size_t)4 * nws), ss = sc("gn_ss", (size_t)4 * B * 2 * x.C);
    View y = view(sc("gn", (size_t)2 * x.rows * x.C), x.rows, x.C);
    Exec* ex = &e;
    const Ref g = wref(nkey + ".g"), bt = wref(nkey + ".b");
    const int Bc = B, groups = c.groups, f32 = x.es == 4;
    const bool has_raw = raw16 != nullptr;
    const View rv = raw16 ? *raw16 : View();
    if (!f32 && !has_raw && mi355x_sd_groupnorm_act_fits(hw, x.C, groups)) {   // one launch for small (batch, group) chunks (unet.py gnorm)
      emit([=](void* st) {
        return mi355x_sd_groupnorm_act(ex->at(x.p), Bc, hw, x.C, x.ld, groups, eps, (const float*)ex->at(g), (const float*)ex->at(bt),
                                       silu ? 1 : 0, ex->at(y.p), y.ld, st);
      });
      return y;
    }
    emit([=](void* st) {
      return mi355x_sd_groupnorm_stats_ex(ex->at(x.p), Bc, hw, x.C, x.ld, groups, eps, (const float*)ex->at(g), (const float*)ex->at(bt),
                                          (float*)ex->at(ws), (float*)ex->at(ss), f32, st);
    });
    emit([=](void* st) {
      return mi355x_sd_scale_shift_act_ex(ex->at(x.p), Bc, hw, x.C, x.ld, (const float*)ex->at(ss), silu ? 1 : 0, ex->at(y.p), y.ld, f32,
                                          has_raw ? ex->at(rv.p) : nullptr, has_raw ? rv.ld : 0, st);
    });
    return y;
  }
  void lnorm(const View& x, const std::string& nkey, const View& out) {
    Exec* ex = &e;
    const Ref g = wref(nkey + ".g"), bt = wref(nkey + ".b");
    const int f32 = x.es == 4;
    emit([=](void* st) {
      return mi355x_sd_layernorm_ex(ex->at(x.p), x.rows, x.C, x.ld, (const float*)ex->at(g), (const float*)ex->at(bt), 1e-5f, ex->at(out.p),
                                    out.ld, f32, st);
    });
  }
  View cast16(const View& x, const std::string& name) {
    if (x.es == 2) return x;
    View y = view(sc(name, (size_t)2 * x.rows * x.C), x.rows, x.C);
    Exec* ex = &e;
    emit([=](void* st) { return mi355x_sd_cast_rows((const float*)ex->at(x.p), x.ld, ex->at(y.p), y.ld, x.rows, x.C, st); });
    return y;
  }
  // bias: additive key mask [B, skv] (fp32), broadcast over heads and queries; has_bias selects the masked kernel
  void attention(const View& q, const View& k, const View& v, const View& out, int heads, int sq, int skv, bool log2 = false,
                 bool has_bias = false, Ref bias = Ref()) {
    const int d = q.C / heads;
    Exec* ex = &e;
    const int Bc = B;
    if (has_bias) {
      // (q may carry head_dim^-0.5 * log2(e) already -- folded into to_q at load: the masked kernel exponentiates
      // exp2((q.k + bias / scale) * scale * log2(e)), so scale = ln 2 makes that exp2(q.k + bias * log2(e)), the same softmax;
      // paddlemix_amd/unet.py attention())
      const float scale = log2 ? (float)log(2.0) : (float)pow((double)d, -0.5);
      emit([=](void* st) {
        return mi355x_sd_sdpa(ex->at(q.p), ex->at(k.p), ex->at(v.p), (const float*)ex->at(bias), ex->at(out.p), Bc, heads, sq, skv, d,
                              (int64_t)sq * q.ld, q.ld, (int64_t)skv * k.ld, k.ld, (int64_t)skv * v.ld, v.ld, (int64_t)sq * out.ld, out.ld,
                              (int64_t)skv, 0, 0, scale, st);
      });
      return;
    }
    if (log2) {
      emit([=](void* st) {
        return mi355x_sd_sdpa_ex(ex->at(q.p), ex->at(k.p), ex->at(v.p), nullptr, ex->at(out.p), Bc, heads, sq, skv, d, (int64_t)sq * q.ld, q.ld,
                                 (int64_t)skv * k.ld, k.ld, (int64_t)skv * v.ld, v.ld, (int64_t)sq * out.ld, out.ld, 0, 0, 0, 1.0f,
                                 MI355X_SD_SDPA_LOG2, st);
      });
      return;
    }
    const float scale = (float)pow((double)d, -0.5);   // == Python d ** -0.5 rounded to fp32
    emit([=](void* st) {
      return mi355x_sd_sdpa(ex->at(q.p), ex->at(k.p), ex->at(v.p), nullptr, ex->at(out.p), Bc, heads, sq, skv, d, (int64_t)sq * q.ld, q.ld,
                            (int64_t)skv * k.ld, k.ld, (int64_t)skv * v.ld, v.ld, (int64_t)sq * out.ld, out.ld, 0, 0, 0, scale, st);
    });
  }

  void build() {
    const int ted = c.boc[0] * 4;
    const int dx = c.cross_dim;
    Exec* ex = &e;
    // ---- inputs ----
    e.in_sample = persist((size_t)4 * B * c.in_channels * H * W);
    e.in_t = persist(4);
    e.in_scale = persist(4);
    e.in_enc = persist((size_t)2 * B * L * dx);
    const View enc = view(e.in_enc, B * L, dx);
    e.out = persist((size_t)4 * B * c.out_channels * H * W);
    // ---- time / added-condition embedding (unet_2d_condition.py:933-1030) ----
    const View t0 = view(persist((size_t)2 * B * c.boc[0]), B, c.boc[0]);
    {
      const Ref tin = e.in_t;
      const int Bc = B, dim = c.boc[0], flip = c.flip_sin_to_cos ? 1 : 0;
      const float fs = (float)c.freq_shift;
      emit([=](void* st) {
        return mi355x_sd_timestep_embedding((const float*)ex->at(tin), 1, Bc, dim, 1, flip, fs, 1.0f, 10000.0f, ex->at(t0.p), dim, st);
      });
    }
    const View e1 = view(persist((size_t)2 * B * ted), B, ted);
    const int ted_b = ted * (c.concat ? 2 : 1);
    const View emb_t = view(persist((size_t)2 * B * ted_b), B, ted_b);
    const View emb = emb_t.cols(0, ted);                   // the time-embedding half (all of it without concat)
    const View cls_out = emb_t.cols(c.concat ? ted : 0, ted);   // where the class embedding goes when concatenated
    // class embedding (unet_2d_condition.py:953-975; same launches as paddlemix_amd/unet.py): a gathered / identity embedding exists
    // before the time MLP and rides in as the residual of its second GEMM; the computed ones add into emb afterwards
    View pre_cls;
    bool has_pre = false;
    if (c.class_kind == Cfg::CLASS_TABLE) {
      e.in_class = persist((size_t)4 * B);
      pre_cls = c.concat ? cls_out : view(persist((size_t)2 * B * ted), B, ted);
      has_pre = !c.concat;
      const Ref ids = e.in_class, tab = wref("class_embedding.table");
      const int Bc = B;
      const View dst = pre_cls;
      emit([=](void* st) { return mi355x_sd_embed_tokens((const int32_t*)ex->at(ids), Bc, 1, ex->at(tab), nullptr, ted, ex->at(dst.p), dst.ld, st); });
    } else if (c.class_kind == Cfg::CLASS_IDENTITY) {
      e.in_class = persist((size_t)2 * B * ted);
      pre_cls = view(e.in_class, B, ted);
      has_pre = !c.concat;
      if (c.concat) {
        const View src = pre_cls, dst = cls_out;
        const int Bc = B;
        emit([=](void* st) { return mi355x_sd_copy_rows(ex->at(src.p), src.ld, ex->at(dst.p), dst.ld, Bc, ted, st); });
      }
    }
    if (c.tcp) {   // t_emb += cond_proj(timestep_cond) (embeddings.py:284-285; zeros when the caller passes none)
      e.in_tcond = persist((size_t)2 * B * c.tcp);
      linear(view(e.in_tcond, B, c.tcp), "time_embedding.cond_proj", t0, false, &t0);
    }
    linear(t0, "time_embedding.linear_1", e1, true, nullptr, MI355X_SD_SILU);
    linear(e1, "time_embedding.linear_2", emb, true, has_pre ? &pre_cls : nullptr);
    if (c.class_kind == Cfg::CLASS_TIMESTEP || c.class_kind == Cfg::CLASS_PROJECTION || c.class_kind == Cfg::CLASS_SIMPLE) {
      View cin;
      if (c.class_kind == Cfg::CLASS_TIMESTEP) {   // class_labels -> sinusoid (time_proj) -> TimestepEmbedding
        e.in_class = persist((size_t)4 * B);
        cin = view(persist((size_t)2 * B * c.boc[0]), B, c.boc[0]);
        const Ref lab = e.in_class;
        const int Bc = B, dim = c.boc[0], flip = c.flip_sin_to_cos ? 1 : 0;
        const float fs = (float)c.freq_shift;
        const View cv = cin;
        emit([=](void* st) {
          return mi355x_sd_timestep_embedding((const float*)ex->at(lab), Bc, Bc, dim, 1, flip, fs, 1.0f, 10000.0f, ex->at(cv.p), cv.ld, st);
        });
      } else {
        e.in_class = persist((size_t)2 * B * c.pdim);
        cin = view(e.in_class, B, c.pdim);
      }
      const View dst = c.concat ? cls_out : emb;
      const View* res = c.concat ? nullptr : &emb;
      if (c.class_kind == Cfg::CLASS_SIMPLE) {
        linear(cin, "class_embedding", dst, true, res);
      } else {
        const View c1 = view(persist((size_t)2 * B * ted), B, ted);
        linear(cin, "class_embedding.linear_1", c1, true, nullptr, MI355X_SD_SILU);
        linear(c1, "class_embedding.linear_2", dst, true, res);
      }
    }
    if (c.text_time) {
      e.text_dim = c.pdim - 6 * c.atd;
      e.n_ids = 6;
      if (e.text_dim <= 0 || (e.text_dim & 7)) die(MI355X_SD_ERR_UNSUPPORTED, "text_time: text_embeds width must be a positive multiple of 8");
      e.in_addin = persist((size_t)2 * B * c.pdim);
      e.in_tids = persist((size_t)4 * B * e.n_ids);
      const View a1 = view(persist((size_t)2 * B * ted), B, ted);
      {
        const Ref tids = e.in_tids, addin = e.in_addin;
        const int n = B * e.n_ids, atd = c.atd, nid = e.n_ids, flip = c.flip_sin_to_cos ? 1 : 0, pdim = c.pdim, td = e.text_dim;
        const float fs = (float)c.freq_shift;
        emit([=](void* st) {
          return mi355x_sd_timestep_embedding((const float*)ex->at(tids), n, n, atd, nid, flip, fs, 1.0f, 10000.0f,
                                              (unsigned char*)ex->at(addin) + 2 * td, pdim, st);
        });
      }
      linear(view(e.in_addin, B, c.pdim), "add_embedding.linear_1", a1, true, nullptr, MI355X_SD_SILU);
      linear(a1, "add_embedding.linear_2", emb, true, &emb);
    }
    const View semb = view(persist((size_t)2 * B * ted_b), B, ted_b);
    {
      const int n = B * ted_b;
      emit([=](void* st) { return mi355x_sd_silu(ex->at(emb_t.p), ex->at(semb.p), n, 0, 0, st); });
    }
    const Ref temb_all = persist((size_t)4 * B * e.temb_total);
    {
      View ta = view(temb_all, B, e.temb_total, 4);
      linear(semb, "temb_all", ta);
    }
    const View kv_all = view(persist((size_t)2 * B * L * e.kv_total), B * L, e.kv_total);
    linear(enc, "kv_all", kv_all, false);
    // IP-Adapter (unet_2d_condition.py:1054-1061): image_embeds -> ImageProjection (Linear, view [B*T, Dx], LayerNorm) -> the image
    // tokens' K/V for every cross-attention in one GEMM; each attn2 then adds scale * attention(q, k_ip, v_ip) (same launches as unet.py)
    View kvip_all;
    const int T = c.ip_tokens;
    if (c.ip) {
      e.in_image = persist((size_t)2 * B * c.ehd);
      const View raw = view(persist((size_t)2 * B * T * dx), B, T * dx);
      const View tok = view(persist((size_t)2 * B * T * dx), B * T, dx);
      linear(view(e.in_image, B, c.ehd), "encoder_hid_proj.image_embeds", raw);
      lnorm(view(raw.p, B * T, dx), "encoder_hid_proj.norm", tok);
      kvip_all = view(persist((size_t)2 * B * T * e.kv_total), B * T, e.kv_total);
      linear(tok, "kvip_all", kvip_all, false);
    }
    const bool ip_on = c.ip && e.ip_scale != 0.0f;
    const float ip_scale = e.ip_scale;

    // ---- optional inputs of this plan (mi355x_sd_unet_plan_ex) ----
    const bool enc_masked = (e.plan_flags & MI355X_SD_UNET_ENC_MASK) != 0, self_masked = (e.plan_flags & MI355X_SD_UNET_SELF_MASK) != 0;
    const bool controlnet = (e.plan_flags & MI355X_SD_UNET_CONTROLNET) != 0;
    if (enc_masked) {
      e.enc_mask = persist((size_t)4 * B * L);
      e.enc_bias = persist((size_t)4 * B * L);
      const Ref m = e.enc_mask, bz = e.enc_bias;
      const int64_t n = (int64_t)B * L;
      emit([=](void* st) { return mi355x_sd_mask_to_bias((const float*)ex->at(m), (float*)ex->at(bz), n, st); });
    }
    if (self_masked) {
      e.self_mask = persist((size_t)4 * B * H * W);
      e.self_bias = persist((size_t)4 * B * H * W);
      const Ref m = e.self_mask, bz = e.self_bias;
      const int64_t n = (int64_t)B * H * W;
      emit([=](void* st) { return mi355x_sd_mask_to_bias((const float*)ex->at(m), (float*)ex->at(bz), n, st); });
    }

    // ---- skip / concat buffers: pre-walk ----
    struct Sk { int C, h, w; };
    std::vector<Sk> skips;
    {
      int h = H, w_ = W, c_cur = c.boc[0];
      for (auto& d : e.layers) {
        if (d.kind == Layer::RESNET) c_cur = d.cout;
        else if (d.kind == Layer::DOWN) { h = (h + 2 - 3) / 2 + 1; w_ = (w_ + 2 - 3) / 2 + 1; }
        else if (d.kind == Layer::SKIP) skips.push_back({c_cur, h, w_});
        else if (d.kind == Layer::CAT) break;
      }
    }
    std::vector<const Layer*> ups;
    for (auto& d : e.layers)
      if (d.kind == Layer::RESNET && d.name.rfind("up_blocks.", 0) == 0) ups.push_back(&d);
    if (ups.size() != skips.size()) die(MI355X_SD_ERR_INVALID, "internal: skip / up-resnet count mismatch");
    std::vector<View> cats;
    std::vector<int> cat_xc;
    for (size_t u = 0; u < ups.size(); ++u) {
      const Sk& s = skips[skips.size() - 1 - u];
      const int cx = ups[u]->cin - s.C;
      cats.push_back(view(persist((size_t)RES * B * s.h * s.w * (cx + s.C)), B * s.h * s.w, cx + s.C, RES));
      cat_xc.push_back(cx);
    }
    auto skip_slot = [&](int k) {
      const size_t u = skips.size() - 1 - k;
      return cats[u].cols(cat_xc[u], cats[u].C - cat_xc[u]);
    };
    auto x_slot = [&](int u) { return cats[u].cols(0, cat_xc[u]); };

    // ---- layer emitters ----
    auto resnet = [&](const Layer& d, const View& x, int h, int w_, const View& out) {
      const int hw = h * w_, rows = B * hw, cout = d.cout;
      const bool need_short = x.C != cout;
      View x16;
      const bool has_x16 = need_short && x.es == 4;
      if (has_x16) x16 = view(sc("x16", (size_t)2 * rows * x.C), rows, x.C);
      const View g1 = gnorm(x, hw, d.name + ".norm1", (float)c.norm_eps, true, has_x16 ? &x16 : nullptr);
      const View h1 = view(sc("h1", (size_t)RES * rows * cout), rows, cout, RES);
      Ref rb = temb_all;
      rb.off += (size_t)4 * e.temb_off[d.name];
      conv3(g1, h, w_, d.name + ".conv1", h1, 1, 0, rb);
      const View g2 = gnorm(h1, hw, d.name + ".norm2", (float)c.norm_eps, true);
      View shortv = x;
      if (need_short) {
        shortv = view(sc("short", (size_t)RES * rows * cout), rows, cout, RES);
        linear(has_x16 ? x16 : x, d.name + ".conv_shortcut", shortv);
      }
      conv3(g2, h, w_, d.name + ".conv2", out, 1, 0, Ref(), &shortv, (float)(1.0 / d.scale));
    };
    auto transformer = [&](const Layer& d, const View& x, int h, int w_, const View& out) {
      const int hw = h * w_, rows = B * hw, ch = x.C;
      const View g = gnorm(x, hw, d.name + ".norm", 1e-6f, false);
      const View hid = view(sc("t_h", (size_t)RES * rows * ch), rows, ch, RES);
      const View hid16 = RES == 4 ? view(sc("t_h16", (size_t)2 * rows * ch), rows, ch) : hid;
      linear(g, d.name + ".proj_in", hid);
      const View ln = view(sc("t_ln", (size_t)2 * rows * ch), rows, ch);
      const View qkv = view(sc("t_qkv", (size_t)2 * rows * 3 * ch), rows, 3 * ch);
      const View ao = view(sc("t_ao", (size_t)2 * rows * ch), rows, ch);
      const View ff = view(sc("t_ff", (size_t)2 * rows * 4 * ch), rows, 4 * ch);
      for (int l = 0; l < d.layers; ++l) {
        const std::string b = d.name + ".transformer_blocks." + std::to_string(l);
        View q2 = view(qkv.p, rows, ch);
        lnorm(hid, b + ".norm1", ln);
        linear(ln, b + ".attn1.qkv", qkv, false);
        if (self_masked && hw != H * W)
          // the reference pads a mask of the wrong length and then fails on the shapes of the add (attention_processor.py:616-622)
          die(MI355X_SD_ERR_INVALID, "attention_mask has " + std::to_string(H * W) + " key tokens but " + b + ".attn1 attends over " +
                                         std::to_string(hw) + " latent tokens (the mask must match the token count of every attention level)");
        attention(qkv.cols(0, ch), qkv.cols(ch, ch), qkv.cols(2 * ch, ch), ao, d.heads, hw, hw, e.log2_blocks.count(b) != 0, self_masked,
                  e.self_bias);
        linear(ao, b + ".attn1.out", hid, true, &hid);
        lnorm(hid, b + ".norm2", ln);
        linear(ln, b + ".attn2.q", q2, false);
        const int ko = e.kv_off[b];
        attention(q2, kv_all.cols(ko, ch), kv_all.cols(ko + ch, ch), ao, d.heads, hw, L, false, enc_masked, e.enc_bias);
        if (ip_on) {   // out += scale * attention over the image tokens (IPAdapterAttnProcessor.__call__, attention_processor.py:1880-1900)
          const View qv = q2, kv = kvip_all.cols(ko, ch), vv = kvip_all.cols(ko + ch, ch), ov = ao;
          const int Bc = B, heads = d.heads, dh = ch / d.heads, Tc = T;
          const float sc_ = (float)pow((double)dh, -0.5);
          emit([=](void* st) {
            return mi355x_sd_sdpa_accum(ex->at(qv.p), ex->at(kv.p), ex->at(vv.p), nullptr, ex->at(ov.p), Bc, heads, hw, Tc, dh, (int64_t)hw * qv.ld,
                                        qv.ld, (int64_t)Tc * kv.ld, kv.ld, (int64_t)Tc * vv.ld, vv.ld, (int64_t)hw * ov.ld, ov.ld, 0, 0, 0, sc_,
                                        ip_scale, st);
          });
        }
        linear(ao, b + ".attn2.out", hid, true, &hid);
        lnorm(hid, b + ".norm3", ln);
        linear(ln, b + ".ff1", ff, true, nullptr, MI355X_SD_GEGLU);
        linear(ff, b + ".ff2", l == d.layers - 1 ? hid16 : hid, true, &hid);
      }
      linear(hid16, d.name + ".proj_out", out, true, &x);
    };

    // ---- body ----
    int h = H, w_ = W, k = 0, u = 0;
    View cur = skip_slot(0);
    {
      const Ref smp = e.in_sample, scl = e.in_scale, wr = wref("conv_in.w"), br = wref("conv_in.b");
      const int Bc = B, ci = c.in_channels, Hc = H, Wc = W, co = c.boc[0], f32 = cur.es == 4;
      const View cv = cur;
      emit([=](void* st) {
        return mi355x_sd_conv_in3x3_ex((const float*)ex->at(smp), (const float*)ex->at(scl), ex->at(wr), (const float*)ex->at(br), ex->at(cv.p),
                                       Bc, ci, Hc, Wc, co, cv.ld, f32, st);
      });
    }
    k = 1;
    int tmp_i = 0;
    auto tmp = [&](int rows, int ch) {
      tmp_i ^= 1;
      return view(sc(std::string("x") + std::to_string(tmp_i), (size_t)RES * rows * ch), rows, ch, RES);
    };
    const size_t nl = e.layers.size();
    for (size_t i = 1; i < nl; ++i) {
      const Layer& d = e.layers[i];
      const Layer::Kind nxt = i + 1 < nl ? e.layers[i + 1].kind : Layer::SKIP;
      const bool last = i + 1 >= nl;
      const int rows = B * h * w_;
      if (d.kind == Layer::RESNET || d.kind == Layer::ATTN) {
        const int cout = d.cout;
        View dst;
        if (!last && nxt == Layer::ATTN) dst = tmp(rows, cout);
        else if (!last && nxt == Layer::SKIP) dst = skip_slot(k);
        else if (!last && nxt == Layer::CAT) dst = x_slot(u);
        else dst = tmp(rows, cout);
        if (d.kind == Layer::RESNET) resnet(d, cur, h, w_, dst);
        else transformer(d, cur, h, w_, dst);
        cur = dst;
      } else if (d.kind == Layer::SKIP) {
        ++k;
      } else if (d.kind == Layer::DOWN) {
        const int ho = (h + 2 - 3) / 2 + 1, wo = (w_ + 2 - 3) / 2 + 1;
        const View dst = skip_slot(k);
        conv3(cast16(cur, "xc16"), h, w_, d.name, dst, 2);
        h = ho;
        w_ = wo;
        cur = dst;
      } else if (d.kind == Layer::CAT) {
        if (u == 0) {
          // every skip tensor and the mid output are complete here: record their shapes (mi355x_sd_unet_skip_shape) and, with
          // ControlNet residuals planned, add them in place (unet_2d_condition.py:1121-1132, 1151-1155) once the down path and the
          // mid block have consumed the originals
          e.skip_shapes.clear();
          for (auto& sk : skips) e.skip_shapes.push_back({sk.C, sk.h, sk.w});
          e.skip_shapes.push_back({cur.C, h, w_});
          if (controlnet) {
            const int f32 = RES == 4 ? 1 : 0, Bc = B;
            e.ctrl_down.clear();
            for (size_t kk = 0; kk < skips.size(); ++kk) {
              const Ref rt = persist((size_t)4 * B * skips[kk].C * skips[kk].h * skips[kk].w);
              e.ctrl_down.push_back(rt);
              const View sl = skip_slot((int)kk);
              const int cs = skips[kk].C;
              const int64_t hw = (int64_t)skips[kk].h * skips[kk].w;
              emit([=](void* st) { return mi355x_sd_add_nchw_ex(ex->at(sl.p), sl.ld, (const float*)ex->at(rt), Bc, cs, hw, f32, st); });
            }
            e.ctrl_mid = persist((size_t)4 * B * cur.C * h * w_);
            const Ref rm = e.ctrl_mid;
            const View cm = cur;
            const int64_t hw = (int64_t)h * w_;
            emit([=](void* st) { return mi355x_sd_add_nchw_ex(ex->at(cm.p), cm.ld, (const float*)ex->at(rm), Bc, cm.C, hw, f32, st); });
          }
        }
        cur = cats[u];
        ++u;
      } else if (d.kind == Layer::UP) {
        const View dst = x_slot(u);
        const Sk& tgt = skips[skips.size() - 1 - u];   // the skip the next resnet concatenates with: the size to reach
        const int ht = tgt.h, wt = tgt.w;
        const View src = cast16(cur, "xc16");
        if (ht == 2 * h && wt == 2 * w_) {
          conv3(src, h, w_, d.name, dst, 1, 1);          // nearest x2 folded into the conv's gather
        } else {
          // `forward_upsample_size` (unet_2d_condition.py:900-906, 1165-1169; Upsample2D interpolates to the skip's size): latents
          // that are not multiples of 2^(levels - 1) leave a skip of odd size 2h - 1, and nearest interpolation from h to 2h - 1 reads
          // source row y >> 1 like the x2 form, cropped. The crop moves the conv's zero padding, so the upsampled tensor is
          // materialised (strided row copies, one per source row and parity: many small launches, on this rare path only) and a
          // plain 3x3 conv follows -- the same launches as paddlemix_amd/unet.py.
          if (!((ht == 2 * h - 1 || ht == 2 * h) && (wt == 2 * w_ - 1 || wt == 2 * w_))) die(MI355X_SD_ERR_INVALID, "internal: upsample target size");
          const int C_ = src.C;
          const View upb = view(sc("upx", (size_t)2 * B * ht * wt * C_), B * ht * wt, C_);
          for (int b = 0; b < B; ++b)
            for (int i_ = 0; i_ < h; ++i_)
              for (int dy = 0; dy < 2; ++dy) {
                const int y = 2 * i_ + dy;
                if (y >= ht) continue;
                for (int dx = 0; dx < 2; ++dx) {
                  const int n = (wt - dx + 1) / 2;
                  Ref sp = src.p, dp = upb.p;
                  sp.off += (size_t)2 * ((size_t)(b * h + i_) * w_) * src.ld;
                  dp.off += (size_t)2 * ((size_t)(b * ht + y) * wt + dx) * C_;
                  const int lds = src.ld;
                  emit([=](void* st) { return mi355x_sd_copy_rows(ex->at(sp), lds, ex->at(dp), 2 * C_, n, C_, st); });
                }
              }
          conv3(upb, ht, wt, d.name, dst);
        }
        h = ht;
        w_ = wt;
        cur = dst;
      }
    }
    if (k != (int)skips.size() || u != (int)ups.size() || h != H || w_ != W) die(MI355X_SD_ERR_INVALID, "internal: plan walk mismatch");
    // ---- post (unet_2d_condition.py:1193-1196) ----
    {
      const View g = gnorm(cur, H * W, "conv_norm_out", (float)c.norm_eps, true);
      const Ref wr = wref("conv_out.w"), br = wref("conv_out.b"), o = e.out;
      const int Bc = B, Hc = H, Wc = W, co = c.out_channels;
      emit([=](void* st) {
        return mi355x_sd_conv_out3x3(ex->at(g.p), g.ld, ex->at(wr), (const float*)ex->at(br), (float*)ex->at(o), Bc, g.C, Hc, Wc, co, st);
      });
    }
    // ---- workspace layout ----
    size_t off = 0;
    for (auto& b : e.bufs) {
      b.off = off;
      off += (b.bytes + 255) & ~(size_t)255;
    }
    e.workspace_bytes = off;
  }
};

thread_local std::string g_exec_err;
int report(const ExecError& x) {
  sd::set_last_error(x.msg.c_str());
  return x.code;
}
Exec* H_(void* h) { return reinterpret_cast<Exec*>(h); }

}  // namespace

extern "C" {

int mi355x_sd_unet_create(const char* config_json, void** handle) {
  if (!config_json || !handle) {
    sd::set_last_error("mi355x_sd_unet_create: null pointer");
    return MI355X_SD_ERR_INVALID;
  }
  try {
    std::unique_ptr<Exec> e(new Exec);
    e->cfg = parse_config(config_json);
    e->layers = structure(e->cfg);
    build_param_table(*e);
    *handle = e.release();
    return MI355X_SD_OK;
  } catch (const ExecError& x) {
    return report(x);
  } catch (const std::exception& x) {   // bad_alloc, a malformed number in the config ...: never across the C boundary
    sd::set_last_error((std::string("mi355x_sd_unet: ") + x.what()).c_str());
    return MI355X_SD_ERR_INVALID;
  }
}

int mi355x_sd_unet_destroy(void* handle) {
  if (!handle) return MI355X_SD_OK;
  Exec* e = H_(handle);
  if (e->graph) (void)hipGraphExecDestroy(e->graph);
  delete e;
  return MI355X_SD_OK;
}

int mi355x_sd_unet_set_option(void* handle, const char* key, int value) {
  if (!handle || !key) {
    sd::set_last_error("mi355x_sd_unet_set_option: null pointer");
    return MI355X_SD_ERR_INVALID;
  }
  Exec* e = H_(handle);
  if (!strcmp(key, "residual_f32")) {
    if (e->planned) {
      sd::set_last_error("mi355x_sd_unet_set_option: residual_f32 must be set before mi355x_sd_unet_plan");
      return MI355X_SD_ERR_INVALID;
    }
    e->resid_f32 = value != 0;
    return MI355X_SD_OK;
  }
  if (!strcmp(key, "fold_softmax_scale")) {
    if (e->packed || e->dev_w) {
      sd::set_last_error("mi355x_sd_unet_set_option: fold_softmax_scale must be set before the weights are packed");
      return MI355X_SD_ERR_INVALID;
    }
    e->fold_scale = value != 0;
    return MI355X_SD_OK;
  }
  sd::set_last_error("mi355x_sd_unet_set_option: unknown option");
  return MI355X_SD_ERR_INVALID;
}

int mi355x_sd_unet_set_ip_adapter_scale(void* handle, float scale) {
  Exec* e = H_(handle);
  if (!handle || !e->cfg.ip) {
    sd::set_last_error("mi355x_sd_unet_set_ip_adapter_scale: this UNet has no IP-Adapter (config encoder_hid_dim_type != 'ip_image_proj')");
    return MI355X_SD_ERR_INVALID;
  }
  if (!(scale == scale) || scale - scale != 0.0f) {
    sd::set_last_error("mi355x_sd_unet_set_ip_adapter_scale: scale must be finite");
    return MI355X_SD_ERR_INVALID;
  }
  if (scale != e->ip_scale && e->planned) {   // the scale is a constant of the planned launches: the plan (and its graph) go
    if (e->graph) {
      (void)hipGraphExecDestroy(e->graph);
      e->graph = nullptr;
    }
    e->drop_plan();
  }
  e->ip_scale = scale;
  return MI355X_SD_OK;
}

int mi355x_sd_unet_num_params(void* handle) { return handle ? (int)H_(handle)->order.size() : -1; }

int mi355x_sd_unet_param_info(void* handle, int index, const char** name, int64_t* shape4, int* ndim) {
  if (!handle || index < 0 || index >= (int)H_(handle)->order.size() || !name || !shape4 || !ndim) {
    sd::set_last_error("mi355x_sd_unet_param_info: bad argument");
    return MI355X_SD_ERR_INVALID;
  }
  Exec* e = H_(handle);
  const std::string& n = e->order[index];
  const HostT& t = e->params[n];
  *name = n.c_str();
  *ndim = (int)t.shape.size();
  for (int i = 0; i < 4; ++i) shape4[i] = i < *ndim ? t.shape[i] : 1;
  return MI355X_SD_OK;
}

int mi355x_sd_unet_load_weight(void* handle, const char* name, const void* host_ptr, const int64_t* shape, int ndim, int dtype) {
  if (!handle || !name || !host_ptr || !shape) {
    sd::set_last_error("mi355x_sd_unet_load_weight: null pointer");
    return MI355X_SD_ERR_INVALID;
  }
  try {
    Exec* e = H_(handle);
    if (e->dev_w || e->packed) die(MI355X_SD_ERR_INVALID, "mi355x_sd_unet_load_weight: weights are already packed / finalized");
    auto it = e->params.find(name);
    if (it == e->params.end()) die(MI355X_SD_ERR_INVALID, std::string("mi355x_sd_unet_load_weight: the configured UNet has no parameter ") + name);
    HostT& t = it->second;
    bool same = (int)t.shape.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = t.shape[i] == shape[i];
    if (!same) die(MI355X_SD_ERR_INVALID, std::string("mi355x_sd_unet_load_weight: ") + name + ": shape differs from the configured model (Paddle layouts expected)");
    const int64_t n = t.numel();
    t.v.resize(n);
    if (dtype == MI355X_SD_DTYPE_F32) {
      memcpy(t.v.data(), host_ptr, n * 4);
    } else if (dtype == MI355X_SD_DTYPE_BF16) {
      const uint16_t* s = (const uint16_t*)host_ptr;
      for (int64_t i = 0; i < n; ++i) t.v[i] = bf16_bits_to_f32(s[i]);
    } else if (dtype == MI355X_SD_DTYPE_F16) {
      const uint16_t* s = (const uint16_t*)host_ptr;
      for (int64_t i = 0; i < n; ++i) t.v[i] = f16_bits_to_f32(s[i]);
    } else {
      die(MI355X_SD_ERR_INVALID, "mi355x_sd_unet_load_weight: dtype must be MI355X_SD_DTYPE_F32 / _BF16 / _F16");
    }
    t.loaded = true;
    return MI355X_SD_OK;
  } catch (const ExecError& x) {
    return report(x);
  } catch (const std::exception& x) {   // bad_alloc, a malformed number in the config ...: never across the C boundary
    sd::set_last_error((std::string("mi355x_sd_unet: ") + x.what()).c_str());
    return MI355X_SD_ERR_INVALID;
  }
}

int mi355x_sd_unet_weight_bytes(void* handle, size_t* bytes) {
  if (!handle || !bytes) {
    sd::set_last_error("mi355x_sd_unet_weight_bytes: null pointer");
    return MI355X_SD_ERR_INVALID;
  }
  try {
    Exec* e = H_(handle);
    if (!e->packed) {   // (not "host_pack is empty": attach / finalize release the host image, the size stays known)
      for (auto& n : e->order)
        if (!e->params[n].loaded) die(MI355X_SD_ERR_INVALID, "mi355x_sd_unet_weight_bytes: parameter " + n + " was never loaded");
      Packer(*e).run();
      e->packed = true;
    }
    *bytes = e->weight_bytes;
    return MI355X_SD_OK;
  } catch (const ExecError& x) {
    return report(x);
  } catch (const std::exception& x) {   // bad_alloc, a malformed number in the config ...: never across the C boundary
    sd::set_last_error((std::string("mi355x_sd_unet: ") + x.what()).c_str());
    return MI355X_SD_ERR_INVALID;
  }
}

int mi355x_sd_unet_pack_weights(void* handle, void* host_buffer, size_t bytes) {
  size_t need = 0;
  const int rc = mi355x_sd_unet_weight_bytes(handle, &need);
  if (rc) return rc;
  Exec* e = H_(handle);
  if (e->host_pack.empty()) {
    sd::set_last_error("mi355x_sd_unet_pack_weights: the packed host image was already released (attach / finalize ran)");
    return MI355X_SD_ERR_INVALID;
  }
  if (!host_buffer || bytes < need) {
    sd::set_last_error("mi355x_sd_unet_pack_weights: buffer missing or too small");
    return MI355X_SD_ERR_INVALID;
  }
  memcpy(host_buffer, e->host_pack.data(), need);
  return MI355X_SD_OK;
}

int mi355x_sd_unet_attach_weights(void* handle, void* device_buffer, size_t bytes) {
  size_t need = 0;
  const int rc = mi355x_sd_unet_weight_bytes(handle, &need);
  if (rc) return rc;
  Exec* e = H_(handle);
  if (!device_buffer || bytes < need || (reinterpret_cast<uintptr_t>(device_buffer) & 255)) {
    sd::set_last_error("mi355x_sd_unet_attach_weights: device buffer missing, too small or not 256-byte aligned");
    return MI355X_SD_ERR_INVALID;
  }
  if (e->dev_w != (unsigned char*)device_buffer) {
    // the plan's launch closures and a captured graph hold ABSOLUTE weight addresses: a moved weight buffer invalidates both
    // (mi355x_sd_unet_plan + _bind_workspace have to run again before the next forward)
    if (e->graph) {
      (void)hipGraphExecDestroy(e->graph);
      e->graph = nullptr;
    }
    e->planned = false;
    e->prog_sym.clear();
  }
  e->dev_w = (unsigned char*)device_buffer;
  std::vector<unsigned char>().swap(e->host_pack);
  return MI355X_SD_OK;
}

int mi355x_sd_unet_packed_tensor(void* handle, const char* key, size_t* offset, size_t* bytes, int* rows, int* cols) {
  Exec* e = H_(handle);
  if (!handle || !key || !offset || !bytes || !rows || !cols || e->w.find(key) == e->w.end()) {
    sd::set_last_error("mi355x_sd_unet_packed_tensor: unknown key (pack the weights first)");
    return MI355X_SD_ERR_INVALID;
  }
  const Packed& p = e->w[key];
  *offset = p.off; *bytes = p.bytes; *rows = p.rows; *cols = p.cols;
  return MI355X_SD_OK;
}

int mi355x_sd_unet_finalize_weights(void* handle, void* device_buffer, size_t bytes, void* stream) {
  size_t need = 0;
  const int rc = mi355x_sd_unet_weight_bytes(handle, &need);
  if (rc) return rc;
  Exec* e = H_(handle);
  if (e->host_pack.empty()) {
    sd::set_last_error("mi355x_sd_unet_finalize_weights: the packed host image was already released (attach / finalize ran); "
                       "use mi355x_sd_unet_attach_weights to move to a buffer that already holds the image");
    return MI355X_SD_ERR_INVALID;
  }
  if (!device_buffer || bytes < need || (reinterpret_cast<uintptr_t>(device_buffer) & 255)) {
    sd::set_last_error("mi355x_sd_unet_finalize_weights: device buffer missing, too small or not 256-byte aligned");
    return MI355X_SD_ERR_INVALID;
  }
  if (hipMemcpyAsync(device_buffer, e->host_pack.data(), need, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(stream)) != hipSuccess ||
      hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {
    sd::set_last_error("mi355x_sd_unet_finalize_weights: host-to-device copy failed");
    return MI355X_SD_ERR_HIP;
  }
  return mi355x_sd_unet_attach_weights(handle, device_buffer, bytes);
}

int mi355x_sd_unet_plan(void* handle, int B, int H, int W, int L, size_t* workspace_bytes) {
  return mi355x_sd_unet_plan_ex(handle, B, H, W, L, 0, workspace_bytes);
}

int mi355x_sd_unet_plan_ex(void* handle, int B, int H, int W, int L, int flags, size_t* workspace_bytes) {
  if (!handle || !workspace_bytes || B <= 0 || H <= 0 || W <= 0 || L <= 0 ||
      (flags & ~(MI355X_SD_UNET_ENC_MASK | MI355X_SD_UNET_SELF_MASK | MI355X_SD_UNET_CONTROLNET))) {
    sd::set_last_error("mi355x_sd_unet_plan: bad argument");
    return MI355X_SD_ERR_INVALID;
  }
  try {
    Exec* e = H_(handle);
    if (!e->dev_w) die(MI355X_SD_ERR_INVALID, "mi355x_sd_unet_plan: call mi355x_sd_unet_finalize_weights first");
    if ((H >> (e->cfg.boc.size() - 1)) < 1 || (W >> (e->cfg.boc.size() - 1)) < 1)
      die(MI355X_SD_ERR_INVALID, "mi355x_sd_unet_plan: latents smaller than one pixel at the lowest level");
    if (e->graph) {
      (void)hipGraphExecDestroy(e->graph);
      e->graph = nullptr;
    }
    e->B = B; e->H = H; e->W = W; e->L = L;
    e->plan_flags = flags;
    e->planned = false;
    e->ctrl_down.clear();
    e->bufs.clear();
    e->buf_index.clear();
    e->prog_sym.clear();
    e->ws = nullptr;
    e->consts_set = false;
    Planner(*e).build();
    e->planned = true;
    *workspace_bytes = e->workspace_bytes;
    return MI355X_SD_OK;
  } catch (const ExecError& x) {
    H_(handle)->drop_plan();   // a failed plan leaves no half-built program behind
    return report(x);
  } catch (const std::exception& x) {   // bad_alloc, a malformed number in the config ...: never across the C boundary
    H_(handle)->drop_plan();
    sd::set_last_error((std::string("mi355x_sd_unet: ") + x.what()).c_str());
    return MI355X_SD_ERR_INVALID;
  }
}

int mi355x_sd_unet_bind_workspace(void* handle, void* device_ptr, size_t bytes) {
  Exec* e = H_(handle);
  if (!handle || !e->planned || !device_ptr || bytes < e->workspace_bytes || (reinterpret_cast<uintptr_t>(device_ptr) & 255)) {
    sd::set_last_error("mi355x_sd_unet_bind_workspace: no plan, or buffer missing / too small / not 256-byte aligned");
    return MI355X_SD_ERR_INVALID;
  }
  if (e->graph) {
    (void)hipGraphExecDestroy(e->graph);
    e->graph = nullptr;
  }
  e->ws = (unsigned char*)device_ptr;
  e->consts_set = false;
  return MI355X_SD_OK;
}

int mi355x_sd_unet_num_launches(void* handle) { return handle ? (int)H_(handle)->prog_sym.size() : -1; }

// sample [B,Cin,H,W] fp32 NCHW, timestep: one fp32, encoder_hidden_states [B,L,D] fp32, text_embeds [B,text_dim] fp32, time_ids
// [B,6] fp32, out [B,Cout,H,W] fp32 -- all DEVICE pointers (the boundary never touches host memory per step); in_scale: optional
// device scalar multiplied into the sample (the scheduler's scale_model_input), NULL = 1. use_graph != 0: the program is captured
// into a hipGraph on first use with this stream and replayed afterwards.
int mi355x_sd_unet_num_skips(void* handle) { return (handle && H_(handle)->planned) ? (int)H_(handle)->skip_shapes.size() - 1 : -1; }

int mi355x_sd_unet_skip_shape(void* handle, int index, int* C, int* H, int* W) {
  Exec* e = H_(handle);
  if (!handle || !e->planned || !C || !H || !W || index < 0 || index >= (int)e->skip_shapes.size()) {
    sd::set_last_error("mi355x_sd_unet_skip_shape: no plan, or index out of range");
    return MI355X_SD_ERR_INVALID;
  }
  *C = e->skip_shapes[index].C; *H = e->skip_shapes[index].h; *W = e->skip_shapes[index].w;
  return MI355X_SD_OK;
}

int mi355x_sd_unet_set_input(void* handle, const char* name, const void* device_ptr) {
  Exec* e = H_(handle);
  if (!handle || !name) {
    sd::set_last_error("mi355x_sd_unet_set_input: null handle or name");
    return MI355X_SD_ERR_INVALID;
  }
  const std::string n(name);
  if (n == "class_labels") {
    if (e->cfg.class_kind == Cfg::CLASS_NONE && device_ptr) {
      // (the reference ignores class_labels on a model without a class embedding, unet_2d_condition.py:953; a binding that
      // nothing will ever read is more likely a wrong handle than intent)
      sd::set_last_error("mi355x_sd_unet_set_input: this model has no class embedding");
      return MI355X_SD_ERR_INVALID;
    }
    e->class_ptr = device_ptr;
    return MI355X_SD_OK;
  }
  if (n == "timestep_cond") {
    if (!e->cfg.tcp && device_ptr) {
      sd::set_last_error("mi355x_sd_unet_set_input: timestep_cond was passed but the model has no `time_cond_proj_dim`");
      return MI355X_SD_ERR_INVALID;
    }
    e->tcond_ptr = static_cast<const float*>(device_ptr);
    return MI355X_SD_OK;
  }
  if (n == "image_embeds") {
    if (!e->cfg.ip && device_ptr) {
      sd::set_last_error("mi355x_sd_unet_set_input: this model has no IP-Adapter (config encoder_hid_dim_type != 'ip_image_proj')");
      return MI355X_SD_ERR_INVALID;
    }
    e->image_ptr = static_cast<const float*>(device_ptr);
    return MI355X_SD_OK;
  }
  sd::set_last_error(("mi355x_sd_unet_set_input: unknown input '" + n + "' (class_labels, timestep_cond, image_embeds)").c_str());
  return MI355X_SD_ERR_INVALID;
}

int mi355x_sd_unet_forward(void* handle, void* stream, const float* sample, const float* timestep, const float* encoder_hidden_states,
                           const float* text_embeds, const float* time_ids, const float* in_scale, float* out, int use_graph) {
  return mi355x_sd_unet_forward_ex(handle, stream, sample, timestep, encoder_hidden_states, text_embeds, time_ids, in_scale, nullptr,
                                   nullptr, nullptr, 0, nullptr, out, use_graph);
}

int mi355x_sd_unet_forward_ex(void* handle, void* stream, const float* sample, const float* timestep, const float* encoder_hidden_states,
                              const float* text_embeds, const float* time_ids, const float* in_scale,
                              const float* encoder_attention_mask, const float* attention_mask,
                              const float* const* down_block_additional_residuals, int num_down_residuals,
                              const float* mid_block_additional_residual, float* out, int use_graph) {
  Exec* e = H_(handle);
  if (!handle || !e->planned || !e->ws) {
    sd::set_last_error("mi355x_sd_unet_forward: plan and bind a workspace first");
    return MI355X_SD_ERR_INVALID;
  }
  if (!sample || !timestep || !encoder_hidden_states || !out) {
    sd::set_last_error("mi355x_sd_unet_forward: null pointer");
    return MI355X_SD_ERR_INVALID;
  }
  if (e->cfg.text_time && (!text_embeds || !time_ids)) {
    // the reference raises ValueError here (unet_2d_condition.py:993-1001)
    sd::set_last_error("mi355x_sd_unet_forward: addition_embed_type 'text_time' requires text_embeds and time_ids");
    return MI355X_SD_ERR_INVALID;
  }
  if (e->cfg.ip && !e->image_ptr) {
    // unet_2d_condition.py:1055-1058
    sd::set_last_error("mi355x_sd_unet_forward: encoder_hid_dim_type 'ip_image_proj' requires image_embeds (mi355x_sd_unet_set_input)");
    return MI355X_SD_ERR_INVALID;
  }
  if (e->cfg.class_kind != Cfg::CLASS_NONE && !e->class_ptr) {
    // unet_2d_condition.py:954-955
    sd::set_last_error("mi355x_sd_unet_forward: class_labels should be provided when num_class_embeds > 0 (mi355x_sd_unet_set_input)");
    return MI355X_SD_ERR_INVALID;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const Cfg& c = e->cfg;
  const int B = e->B, H = e->H, W = e->W, L = e->L;
  // ---- the optional inputs must be exactly the ones the plan was built for ----
  const bool want_enc = (e->plan_flags & MI355X_SD_UNET_ENC_MASK) != 0, want_self = (e->plan_flags & MI355X_SD_UNET_SELF_MASK) != 0;
  const bool want_ctrl = (e->plan_flags & MI355X_SD_UNET_CONTROLNET) != 0;
  const bool has_ctrl = down_block_additional_residuals != nullptr || mid_block_additional_residual != nullptr;
  if (want_enc != (encoder_attention_mask != nullptr) || want_self != (attention_mask != nullptr) || want_ctrl != has_ctrl) {
    sd::set_last_error("mi355x_sd_unet_forward_ex: optional inputs differ from the plan's flags (mi355x_sd_unet_plan_ex)");
    return MI355X_SD_ERR_INVALID;
  }
  if (want_ctrl) {
    if (!down_block_additional_residuals || !mid_block_additional_residual) {
      // the reference's T2I-adapter form (down residuals only) is not built
      sd::set_last_error("mi355x_sd_unet_forward_ex: ControlNet residuals need both down_block_additional_residuals and "
                         "mid_block_additional_residual");
      return MI355X_SD_ERR_INVALID;
    }
    if (num_down_residuals != (int)e->ctrl_down.size()) {
      sd::set_last_error("mi355x_sd_unet_forward_ex: wrong number of down_block_additional_residuals (mi355x_sd_unet_num_skips)");
      return MI355X_SD_ERR_INVALID;
    }
    for (int i = 0; i < num_down_residuals; ++i)
      if (!down_block_additional_residuals[i]) {
        sd::set_last_error("mi355x_sd_unet_forward_ex: null residual pointer");
        return MI355X_SD_ERR_INVALID;
      }
  }
  // ---- stage inputs into the plan's static buffers (device-to-device, stream ordered) ----
  bool ok = hipMemcpyAsync(e->at(e->in_sample), sample, (size_t)4 * B * c.in_channels * H * W, hipMemcpyDeviceToDevice, st) == hipSuccess;
  if (want_enc) ok = ok && hipMemcpyAsync(e->at(e->enc_mask), encoder_attention_mask, (size_t)4 * B * L, hipMemcpyDeviceToDevice, st) == hipSuccess;
  if (want_self) ok = ok && hipMemcpyAsync(e->at(e->self_mask), attention_mask, (size_t)4 * B * H * W, hipMemcpyDeviceToDevice, st) == hipSuccess;
  if (want_ctrl) {
    for (size_t i = 0; i < e->ctrl_down.size(); ++i) {
      const Exec::SkipShape& sk = e->skip_shapes[i];
      ok = ok && hipMemcpyAsync(e->at(e->ctrl_down[i]), down_block_additional_residuals[i], (size_t)4 * B * sk.C * sk.h * sk.w,
                                hipMemcpyDeviceToDevice, st) == hipSuccess;
    }
    const Exec::SkipShape& mk = e->skip_shapes.back();
    ok = ok && hipMemcpyAsync(e->at(e->ctrl_mid), mid_block_additional_residual, (size_t)4 * B * mk.C * mk.h * mk.w, hipMemcpyDeviceToDevice,
                              st) == hipSuccess;
  }
  ok = ok && hipMemcpyAsync(e->at(e->in_t), timestep, 4, hipMemcpyDeviceToDevice, st) == hipSuccess;
  if (in_scale) {
    ok = ok && hipMemcpyAsync(e->at(e->in_scale), in_scale, 4, hipMemcpyDeviceToDevice, st) == hipSuccess;
    e->consts_set = false;
  } else if (!e->consts_set) {
    static const float one = 1.0f;
    ok = ok && hipMemcpyAsync(e->at(e->in_scale), &one, 4, hipMemcpyHostToDevice, st) == hipSuccess;
    e->consts_set = true;
  }
  if (!ok) {
    sd::set_last_error("mi355x_sd_unet_forward: staging copy failed");
    return MI355X_SD_ERR_HIP;
  }
  int rc = mi355x_sd_cast_rows(encoder_hidden_states, c.cross_dim, e->at(e->in_enc), c.cross_dim, (int64_t)B * L, c.cross_dim, stream);
  if (rc) return rc;
  if (c.ip) {
    rc = mi355x_sd_cast_rows(e->image_ptr, c.ehd, e->at(e->in_image), c.ehd, B, c.ehd, stream);
    if (rc) return rc;
  }
  if (c.tcp) {
    if (e->tcond_ptr) {
      rc = mi355x_sd_cast_rows(e->tcond_ptr, c.tcp, e->at(e->in_tcond), c.tcp, B, c.tcp, stream);
      if (rc) return rc;
    } else if (hipMemsetAsync(e->at(e->in_tcond), 0, (size_t)2 * B * c.tcp, st) != hipSuccess) {
      sd::set_last_error("mi355x_sd_unet_forward: staging copy failed");
      return MI355X_SD_ERR_HIP;
    }
  }
  if (c.class_kind == Cfg::CLASS_TABLE || c.class_kind == Cfg::CLASS_TIMESTEP) {   // int32 indices / fp32 values, one per sample
    if (hipMemcpyAsync(e->at(e->in_class), e->class_ptr, (size_t)4 * B, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      sd::set_last_error("mi355x_sd_unet_forward: staging copy failed");
      return MI355X_SD_ERR_HIP;
    }
  } else if (c.class_kind != Cfg::CLASS_NONE) {
    const int wdt = c.class_kind == Cfg::CLASS_IDENTITY ? c.boc[0] * 4 : c.pdim;
    rc = mi355x_sd_cast_rows((const float*)e->class_ptr, wdt, e->at(e->in_class), wdt, B, wdt, stream);
    if (rc) return rc;
  }
  if (c.text_time) {
    rc = mi355x_sd_cast_rows(text_embeds, e->text_dim, e->at(e->in_addin), c.pdim, B, e->text_dim, stream);
    if (rc) return rc;
    if (hipMemcpyAsync(e->at(e->in_tids), time_ids, (size_t)4 * B * e->n_ids, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      sd::set_last_error("mi355x_sd_unet_forward: staging copy failed");
      return MI355X_SD_ERR_HIP;
    }
  }
  // ---- the program ----
  auto run_eager = [&]() {
    for (auto& f : e->prog_sym) {
      const int r = f(stream);
      if (r) return r;
    }
    return (int)MI355X_SD_OK;
  };
  if (use_graph) {
    if (!e->graph || e->graph_stream != stream) {
      if (e->graph) {
        (void)hipGraphExecDestroy(e->graph);
        e->graph = nullptr;
      }
      rc = run_eager();   // warm-up outside capture (lazy module loading), also this call's result
      if (rc) return rc;
      rc = mi355x_sd_graph_begin(stream);
      if (rc) return rc;
      const int rprog = run_eager();
      void* exec = nullptr;
      rc = mi355x_sd_graph_end(stream, &exec);
      if (rprog || rc) {   // a launch failed inside the capture: do not keep (or leak) the partial graph
        if (exec) (void)hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(exec));
        return rprog ? rprog : rc;
      }
      e->graph = reinterpret_cast<hipGraphExec_t>(exec);
      e->graph_stream = stream;
    } else {
      rc = mi355x_sd_graph_launch(e->graph, stream);
      if (rc) return rc;
    }
  } else {
    rc = run_eager();
    if (rc) return rc;
  }
  if (hipMemcpyAsync(out, e->at(e->out), (size_t)4 * B * c.out_channels * H * W, hipMemcpyDeviceToDevice, st) != hipSuccess) {
    sd::set_last_error("mi355x_sd_unet_forward: output copy failed");
    return MI355X_SD_ERR_HIP;
  }
  return MI355X_SD_OK;
}

}  // extern "C"
