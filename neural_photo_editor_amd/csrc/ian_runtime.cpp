// libian.so runtime: model handle, parameter folding / repacking, work-table scheduling, executor, C ABI.
// Host-side C++ (compiled by hipcc for the HIP host API); all arithmetic on activations is in the
// kernels_*.hip files.  See include/ian.h for the contract and the reference lines each entry replaces.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ian.h"
#include "../../include/ian_train.h"
#include "ian_internal.h"
#include "ian_guard.h"   // IAN_SANITIZE builds: guard bands around every device allocation (no-op otherwise)

using namespace ian;

namespace {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int ilog2_exact(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

struct Schedule {
  int variant = -1;
  int cfg = 0;
  int nitems = 0;
  TgItem* d_items = nullptr;
  int ntiles = 0;  // >0 => split-K: slabs + reduce pass
  TgTile* d_tiles = nullptr;
  int* d_counters = nullptr;   // split-K combine fused into the tapgemm launch: one arrival counter per tile (zero at rest)
  size_t slab_tiles = 0;
  int max_nsplit = 1;
  std::vector<TgItem> h_items;  // kept for tests / debugging
  std::vector<TgTile> h_tiles;
};

struct TgChoice {  // autotuned (or forced) schedule shape for one (layer, batch)
  int cfg = -1;        // enum TgConfig, -1 = heuristic
  int max_steps = -1;  // -1 = heuristic, 0 = never split K, >0 = split so that no item exceeds this many K-steps
  int variant = -1;    // K-loop schedule of tapgemm_kernel, -1 = the handle's option
};

// one linear map executed by the tapgemm kernel (forward or backward-data form of an op)
struct TgLayer {
  bool valid = false;
  int IH = 1, IW = 1, Cin = 32, QH = 1, QW = 1, si = 1, by = 0, bx = 0, so = 1, OH = 1, OW = 1, Cout = 0, CoutPad = 0;
  int cin_real = 0;  // for FLOP accounting
  std::vector<TgClass> classes;
  std::vector<TgTap> taps;
  std::vector<float> h_w;  // packed weights (freed after upload)
  size_t w_floats = 0;
  float* d_w = nullptr;
  TgClass* d_classes = nullptr;
  TgTap* d_taps = nullptr;
  std::map<int, Schedule> sched;  // per batch size
  std::map<int, TgChoice> choice;  // per batch size, set by ian_autotune
  double macs_per_image() const {
    double m = 0;
    for (auto& c : classes) m += (double)QH * QW * c.ntaps * cin_real * Cout;
    return m;
  }
};

struct Slot {
  int h = 1, w = 1, c = 1, cs = 32;
  bool nchw = false;  // external-layout tensors (image in / image out)
  float* d = nullptr;
  float* g = nullptr;  // gradient wrt the producer's pre-epilogue value (latent-brush backward)
  size_t cap = 0, gcap = 0;
  size_t per_image() const { return nchw ? (size_t)c * h * w : (size_t)h * w * cs; }
};

struct OpPlan {
  ian_op_desc d;
  std::string name, bn_name;
  TgLayer fwd, bwd;
  bool edge = false;  // 3-channel edge kernel instead of tapgemm
  float* d_edge_w = nullptr;
  std::vector<float> h_edge_w;
  float* d_scale = nullptr;
  float* d_shift = nullptr;
  std::vector<float> h_scale, h_shift;
  float* d_made_w = nullptr;
  float* d_made_b = nullptr;
  // batch-1 streaming form of a transposed conv and of its backward-data (kernels_b1.hip)
  std::vector<float> h_b1_fwd, h_b1_bwd;
  float* d_b1_fwd = nullptr;
  float* d_b1_bwd = nullptr;
  long long b1_cls_off[4] = {0, 0, 0, 0};
  long long head_done_serial = -1;  // run_serial of the call in which a sibling's fused head launch produced this op
};

struct Options {
  int tg_cfg = -1;            // force a tile config (enum TgConfig) or -1 = auto
  int tg_target_items = 768;  // split-K aims at about this many workgroups
  int tg_min_steps = 16;      // never make a K-range shorter than this many 32-channel steps
  int tg_no_split_items = 384;  // do not split when tiles alone give at least this many workgroups
  int tg_split = 1;
  int tg_xcd_group = 8;       // supergroup edge (tiles) dealt to one XCD
  int tg_prefer_nosplit = 1;  // try smaller tiles before resorting to split-K
  int tg_nosplit_min_out = 1 << 30;  // outputs (M*Cout) above which 64x64 is forced even if it under-fills
  int mdc_head = 2;                  // few-filter MDCL layers: 0 = tapgemm, 1 = VALU head kernel, 2 = + sibling layers fused
  int tg_variant = 2;                // K-loop schedule of tapgemm_kernel (kernels_tapgemm.hip); autotune picks per layer
  int tg_fused_reduce_max_m = 0;     // split-K combine by the last-arriving workgroup (no reduce launch) when images*QH*QW <= this.
                                     // OFF: measured 3x slower per layer at batch 1 (14 -> 37-48 us) -- the agent-scope release every
                                     // workgroup needs before it bumps the arrival counter is a whole-L2 writeback on gfx950
  int tg_reduce_kp = 4;              // split-K reduce: lanes sharing one output element's slabs when a tile has >= 8 slabs
  int wg_w8 = 1;                     // tapwgrad: 8-wave 128x128 workgroups (16 waves per CU instead of 8)
  int wg_target_items = 1024;        // tapwgrad: split the pixel range until taps x channel tiles x splits reaches this many workgroups
  int mdc_thin_tile = 1;             // thin MDCL (G_b / B_b and their backward-data) with the input rows staged through LDS
  int b1_conv = 0;                   // batch-1 transposed convs and their backward-data as whole-contraction streaming launches
                                     // (kernels_b1.hip).  OFF: measured slower than tapgemm + reduce (brush event 0.178 vs 0.160 ms):
                                     // 16-pixel tiles re-read weights and input rows from L2 at 4 FLOP/B
  int dec_out_wgs = 256;             // image-producing deconv: split images into row bands until this many workgroups exist
  int alias_io = 1;                  // ian_reconstruct: device-pointer images are read / written in place (no boundary copies)
  int dense_gemv = 1;                // batch-1 backward of the dense layer fed by the latent as one GEMV launch
  int dec_out_px = 1;                // ... and, below that batch, 8 lanes per output pixel instead of 16 tile workgroups
  int dec_out_mfma = 1;              // image-producing deconv (IAN_simple dec_out) on the matrix cores for batches >= 4
  int edit_graph = 1;                // batch-1 host-pointer calls (the NPE edit loop) replay captured hipGraphs
  int head_fused = 1;                // RGB-Beta head as head6 + head_tail (kernels_head.hip) when the graph matches IAN.py:183-207
  int head_fused_min_n = 8;          // ... for batches of at least this many images (the latent brush's batch-1 backward
                                     // needs the per-layer activations of the unfused ops)
};

// IAN.py:183-207 recognised in the lowered decoder: op indices of R, G_a, G_b, B_a, B_b, the [R,G] concat and the beta op
struct HeadPlan {
  bool searched = false, valid = false;
  int opR = -1, opGa = -1, opGb = -1, opBa = -1, opBb = -1, opCat = -1, opBeta = -1, first = -1;
  int* d_itab = nullptr;
  float* d_ftab = nullptr;
  float* d_comp = nullptr;
  size_t comp_cap = 0;
  int halo = 0;
};

}  // namespace

struct ian_handle {
  ian_model_desc desc;
  std::vector<OpPlan> ops;
  std::vector<Slot> slots;
  std::vector<std::string> strings;
  std::map<std::string, HostTensor> params;
  std::vector<float> made_masks[3];
  int made_n = 0;
  bool finalized = false;
  Options opt;
  std::string err;
  // workspaces
  float* d_slab = nullptr;
  size_t slab_cap = 0;
  float* d_stage_in = nullptr;
  size_t stage_in_cap = 0;
  float* d_stage_out = nullptr;
  size_t stage_out_cap = 0;
  float* d_gseed = nullptr;  // 3*H*W gradient seed
  float* d_rgb = nullptr;
  // decoder-forward cache for the interactive loop (NPE.py:205,218: imgradRGB(z) right after sample_at(z)):
  // the batch-1 decoder activations of the last HOST latent are kept; a gradient call on the same latent skips
  // its forward pass.  Any other use of the decoder slots invalidates it.
  long long run_serial = 0;  // incremented per executed segment (fused head bookkeeping)
  std::map<int, char> slot_stale;  // external-layout slots whose own buffer was bypassed by a device-pointer call (SlotAlias)
  std::vector<float> dec_cache_z;
  std::vector<float> rgb_cache;  // host copy of the brush image last uploaded to d_rgb
  // NPE.paint photo blend (ian_photo_blend): device copies of RECON / ERROR with their host shadows, outputs
  unsigned char* d_recon = nullptr;
  float* d_error = nullptr;
  unsigned char* d_im = nullptr;
  double* d_mask = nullptr;
  std::vector<unsigned char> recon_cache;
  std::vector<float> error_cache;
  unsigned char* d_u8 = nullptr;
  size_t u8_cap = 0;
  bool dec_cache_valid = false;
  HeadPlan head;
  // Interactive loop (NPE.py:192-235): batch-1 calls with HOST pointers on the default stream run on an internal stream
  // and, from the third call on, replay a captured hipGraph (decoder forward; backward chain per loss kind), removing
  // the ~15 launches x 3-4 us of host launch time per call.  alloc_epoch counts (re)allocations of anything a captured
  // kernel argument may point to; a graph captured under an older epoch is dropped and re-captured.
  struct EditGraph {
    hipGraphExec_t exec = nullptr;
    long long epoch = -1;
    int warm = 0;
    long long warm_epoch = -1;   // the eager pass counts only for the options / schedules / buffers it ran with
  };
  long long alloc_epoch = 0;
  hipStream_t edit_stream = nullptr;
  float* pin = nullptr;      // pinned host block: z [0,128) | dz [128,256) | image [256, 256+12288) | patch (4 ints) after that
  int* d_patch = nullptr;
  EditGraph g_fwd, g_bwd[2], g_step[2][2];   // g_step[mode][image wanted]: backward + latent update + forward (ian_brush_step)
  bool graph_failed = false;
  hipStream_t last_stream = nullptr;   // stream of the last call that left work un-synchronised (or nullptr)
  bool last_pending = false;
  // profiling
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_total;
  size_t ev_total_used = 0;
  double prof_flops = 0;
  int64_t prof_launches = 0;
};

namespace {

// key=value tuning knobs (ian_set_option; also read once from the environment variable IAN_OPTS="k=v,k=v" when a handle
// or a training layer is created, so that tests and profiling runs can pin a schedule policy for every object)
bool apply_option(Options& o, const std::string& k, int value) {
  if (k == "tg_cfg") o.tg_cfg = value;
  else if (k == "tg_target_items") o.tg_target_items = std::max(1, value);
  else if (k == "tg_min_steps") o.tg_min_steps = std::max(1, value);
  else if (k == "tg_no_split_items") o.tg_no_split_items = value;
  else if (k == "tg_split") o.tg_split = value;
  else if (k == "tg_xcd_group") o.tg_xcd_group = std::max(1, value);
  else if (k == "tg_prefer_nosplit") o.tg_prefer_nosplit = value;
  else if (k == "tg_nosplit_min_out") o.tg_nosplit_min_out = value;
  else if (k == "tg_variant") {
#ifdef IAN_ABLATION
    const bool ok = (value >= 0 && value <= 4) || (value >= 10 && value <= 12);
#else
    const bool ok = value >= 0 && value <= 4;   // 10..12 are timing-only ablations with wrong results (-DIAN_ABLATION builds only)
#endif
    if (!ok) return false;
    o.tg_variant = value;
  }
  else if (k == "tg_reduce_kp") o.tg_reduce_kp = value;
  else if (k == "tg_fused_reduce_max_m") o.tg_fused_reduce_max_m = value;
  else if (k == "mdc_head") o.mdc_head = value;
  else if (k == "head_fused") o.head_fused = value;
  else if (k == "head_fused_min_n") o.head_fused_min_n = std::max(1, value);
  else if (k == "edit_graph") o.edit_graph = value;
  else if (k == "dec_out_mfma") o.dec_out_mfma = value;
  else if (k == "dec_out_px") o.dec_out_px = value;
  else if (k == "dense_gemv") o.dense_gemv = value;
  else if (k == "b1_conv") o.b1_conv = value;
  else if (k == "alias_io") o.alias_io = value;
  else if (k == "dec_out_wgs") o.dec_out_wgs = std::max(1, value);
  else if (k == "mdc_thin_tile") o.mdc_thin_tile = value;
  else if (k == "wg_target_items") o.wg_target_items = value;
  else if (k == "wg_w8") o.wg_w8 = value;
  else return false;
  return true;
}
void apply_env_options(Options& o) {
  const char* env = getenv("IAN_OPTS");
  if (!env) return;
  std::string sv(env);
  size_t pos = 0;
  while (pos < sv.size()) {
    size_t end = sv.find(',', pos);
    if (end == std::string::npos) end = sv.size();
    const std::string kv = sv.substr(pos, end - pos);
    const size_t eq = kv.find('=');
    if (eq != std::string::npos) (void)apply_option(o, kv.substr(0, eq), atoi(kv.c_str() + eq + 1));
    pos = end + 1;
  }
}

int fail(ian_handle* h, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  return code;
}

#define HIPCHK(h, expr)                                                                            \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(h, -2, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
  } while (0)

const HostTensor* find_param(ian_handle* h, const std::string& n) {
  auto it = h->params.find(n);
  return it == h->params.end() ? nullptr : &it->second;
}

int need_param(ian_handle* h, const std::string& n, std::initializer_list<int64_t> shape, const HostTensor** out) {
  const HostTensor* t = find_param(h, n);
  if (!t) return fail(h, -3, "missing parameter '%s'", n.c_str());
  std::vector<int64_t> want(shape);
  if (t->shape != want) {
    std::string a, b;
    for (auto s : t->shape) a += std::to_string(s) + ",";
    for (auto s : want) b += std::to_string(s) + ",";
    return fail(h, -3, "parameter '%s' has shape (%s) expected (%s)", n.c_str(), a.c_str(), b.c_str());
  }
  *out = t;
  return 0;
}

template <typename T>
int upload(ian_handle* h, const std::vector<T>& v, T** dptr) {
  if (v.empty()) {
    *dptr = nullptr;
    return 0;
  }
  HIPCHK(h, hipMalloc((void**)dptr, v.size() * sizeof(T)));
  HIPCHK(h, hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  ++h->alloc_epoch;
  return 0;
}

// ----- epilogue vectors: fold BatchNorm (App. B.3) and bias into y = acc*scale + shift --------------
int build_affine(ian_handle* h, OpPlan& op, int features, const std::vector<int>* perm) {
  std::vector<float> sc(features, 1.f), sh(features, 0.f);
  const HostTensor* b = nullptr;
  if (op.d.has_bias) {
    int rc = need_param(h, op.name + ".b", {features}, &b);
    if (rc) return rc;
  }
  if (!op.bn_name.empty()) {
    const HostTensor *g, *be, *m, *is;
    int rc;
    if ((rc = need_param(h, op.bn_name + ".gamma", {features}, &g))) return rc;
    if ((rc = need_param(h, op.bn_name + ".beta", {features}, &be))) return rc;
    if ((rc = need_param(h, op.bn_name + ".mean", {features}, &m))) return rc;
    if ((rc = need_param(h, op.bn_name + ".inv_std", {features}, &is))) return rc;
    for (int i = 0; i < features; ++i) {
      const float s = g->data[i] * is->data[i];
      sc[i] = s;
      // (x + b - mean)*s + beta
      sh[i] = be->data[i] + ((b ? b->data[i] : 0.f) - m->data[i]) * s;
    }
  } else if (b) {
    for (int i = 0; i < features; ++i) sh[i] = b->data[i];
  }
  const int padded = round_up(features, 128);
  op.h_scale.assign(padded, 1.f);
  op.h_shift.assign(padded, 0.f);
  for (int i = 0; i < features; ++i) {
    const int dst = perm ? (*perm)[i] : i;
    op.h_scale[dst] = sc[i];
    op.h_shift[dst] = sh[i];
  }
  return 0;
}

// reference (C,H,W)-flattened index -> internal (H,W,C) index
std::vector<int> chw_to_hwc_perm(int C, int H, int W) {
  std::vector<int> p((size_t)C * H * W);
  for (int c = 0; c < C; ++c)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) p[((size_t)c * H + y) * W + x] = (y * W + x) * C + c;
  return p;
}

void add_class(TgLayer& L, int py, int px, const std::vector<TgTap>& taps, long long w_off) {
  TgClass c;
  c.ntaps = (int)taps.size();
  c.tap0 = (int)L.taps.size();
  c.py = py;
  c.px = px;
  c.w_off = w_off;
  L.classes.push_back(c);
  for (auto& t : taps) L.taps.push_back(t);
}

// ----- weight repacking ------------------------------------------------------------------------------
// CONV5S2 forward: IAN_simple.py:73-116. W (Cout,Cin,5,5); slab t=ky*5+kx: [CoutPad][CinPad] = W[co,ci,ky,kx]
int pack_conv_fwd(ian_handle* h, OpPlan& op, bool allow_edge = true) {
  const int cin = op.d.cin, cout = op.d.cout, H = op.d.in_h, W = op.d.in_w;
  const HostTensor* Wt;
  int rc = need_param(h, op.name + ".W", {cout, cin, 5, 5}, &Wt);
  if (rc) return rc;
  if (cin < 8 && allow_edge) {  // edge kernel: [75][Cout], k=(c*5+ky)*5+kx
    op.edge = true;
    op.h_edge_w.assign((size_t)75 * cout, 0.f);
    for (int co = 0; co < cout; ++co)
      for (int c = 0; c < cin; ++c)
        for (int k = 0; k < 25; ++k) op.h_edge_w[(size_t)(c * 25 + k) * cout + co] = Wt->data[((size_t)co * cin + c) * 25 + k];
    return 0;
  }
  TgLayer& L = op.fwd;
  L.valid = true;
  L.IH = H; L.IW = W; L.Cin = round_up(cin, 32); L.cin_real = cin;
  L.QH = H / 2; L.QW = W / 2; L.si = 2; L.by = -2; L.bx = -2; L.so = 1; L.OH = H / 2; L.OW = W / 2;
  L.Cout = cout; L.CoutPad = round_up(cout, 128);
  std::vector<TgTap> taps;
  for (int ky = 0; ky < 5; ++ky)
    for (int kx = 0; kx < 5; ++kx) taps.push_back({ky, kx});
  add_class(L, 0, 0, taps, 0);
  const size_t slab = (size_t)L.CoutPad * L.Cin;
  L.h_w.assign(slab * 25, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int k = 0; k < 25; ++k) L.h_w[k * slab + (size_t)co * L.Cin + ci] = Wt->data[((size_t)co * cin + ci) * 25 + k];
  return 0;
}

// CONV5S2 backward-data (training): dX[iy,ix,ci] = sum dY[oy,ox,co] * W[co,ci,ky,kx] with iy = 2*oy - 2 + ky.
// Same parity-class decomposition as the transposed conv's forward (gather form, no atomics): class (py,px) =
// parity of (iy,ix), taps ky = py, py+2, ..., oy = qy + (py + 2 - ky)/2.  Slabs [ci][co].
int pack_conv_bwd(ian_handle* h, OpPlan& op) {
  const int cin = op.d.cin, cout = op.d.cout, H = op.d.in_h, W = op.d.in_w;
  const HostTensor* Wt;
  int rc = need_param(h, op.name + ".W", {cout, cin, 5, 5}, &Wt);
  if (rc) return rc;
  TgLayer& L = op.bwd;
  L.valid = true;
  L.IH = H / 2; L.IW = W / 2; L.Cin = round_up(cout, 32); L.cin_real = cout;
  L.QH = H / 2; L.QW = W / 2; L.si = 1; L.by = 0; L.bx = 0; L.so = 2; L.OH = H; L.OW = W;
  L.Cout = cin; L.CoutPad = round_up(cin, 128);
  const size_t slab = (size_t)L.CoutPad * L.Cin;
  L.h_w.assign(slab * 25, 0.f);
  size_t t_global = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      std::vector<TgTap> taps;
      const long long w_off = (long long)(t_global * slab);
      for (int ky = py; ky < 5; ky += 2)
        for (int kx = px; kx < 5; kx += 2) {
          taps.push_back({(py + 2 - ky) / 2, (px + 2 - kx) / 2});
          float* dst = L.h_w.data() + t_global * slab;
          for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) dst[(size_t)ci * L.Cin + co] = Wt->data[(((size_t)co * cin + ci) * 5 + ky) * 5 + kx];
          ++t_global;
        }
      add_class(L, py, px, taps, w_off);
    }
  return 0;
}

// DECONV5S2: layers.py:436-483. W (Cin,Cout,5,5). out[oy] += x[iy]*W[ci,co,kt], oy=2iy-2+ky, kt=flip?4-k:k.
// forward: 4 parity classes; backward-data: 25-tap stride-2 "conv" over dY with [ci][co] slabs.
int pack_deconv(ian_handle* h, OpPlan& op) {
  const int cin = op.d.cin, cout = op.d.cout, H = op.d.in_h, W = op.d.in_w;
  const bool flip = h->desc.deconv_flip != 0;
  const HostTensor* Wt;
  int rc = need_param(h, op.name + ".W", {cin, cout, 5, 5}, &Wt);
  if (rc) return rc;
  auto wref = [&](int ci, int co, int ky, int kx) {
    const int ty = flip ? 4 - ky : ky, tx = flip ? 4 - kx : kx;
    return Wt->data[(((size_t)ci * cout + co) * 5 + ty) * 5 + tx];
  };
  if (cout <= 4) {  // edge kernel: [25][4][Cin]
    op.edge = true;
    op.h_edge_w.assign((size_t)25 * 4 * cin, 0.f);
    for (int ky = 0; ky < 5; ++ky)
      for (int kx = 0; kx < 5; ++kx)
        for (int co = 0; co < cout; ++co)
          for (int ci = 0; ci < cin; ++ci) op.h_edge_w[((size_t)(ky * 5 + kx) * 4 + co) * cin + ci] = wref(ci, co, ky, kx);
    return 0;
  }
  {
    TgLayer& L = op.fwd;
    L.valid = true;
    L.IH = H; L.IW = W; L.Cin = round_up(cin, 32); L.cin_real = cin;
    L.QH = H; L.QW = W; L.si = 1; L.by = 0; L.bx = 0; L.so = 2; L.OH = 2 * H; L.OW = 2 * W;
    L.Cout = cout; L.CoutPad = round_up(cout, 128);
    const size_t slab = (size_t)L.CoutPad * L.Cin;
    L.h_w.assign(slab * 25, 0.f);
    size_t t_global = 0;
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        std::vector<TgTap> taps;
        const long long w_off = (long long)(t_global * slab);
        for (int ky = py; ky < 5; ky += 2)
          for (int kx = px; kx < 5; kx += 2) {
            taps.push_back({(py + 2 - ky) / 2, (px + 2 - kx) / 2});
            float* dst = L.h_w.data() + t_global * slab;
            for (int co = 0; co < cout; ++co)
              for (int ci = 0; ci < cin; ++ci) dst[(size_t)co * L.Cin + ci] = wref(ci, co, ky, kx);
            ++t_global;
          }
        add_class(L, py, px, taps, w_off);
      }
  }
  {
    TgLayer& L = op.bwd;  // dX[iy,ix,ci] = sum dY[2iy-2+ky, 2ix-2+kx, co] * Wt[ci,co,ky,kx]
    L.valid = true;
    L.IH = 2 * H; L.IW = 2 * W; L.Cin = round_up(cout, 32); L.cin_real = cout;
    L.QH = H; L.QW = W; L.si = 2; L.by = -2; L.bx = -2; L.so = 1; L.OH = H; L.OW = W;
    L.Cout = cin; L.CoutPad = round_up(cin, 128);
    std::vector<TgTap> taps;
    for (int ky = 0; ky < 5; ++ky)
      for (int kx = 0; kx < 5; ++kx) taps.push_back({ky, kx});
    add_class(L, 0, 0, taps, 0);
    const size_t slab = (size_t)L.CoutPad * L.Cin;
    L.h_w.assign(slab * 25, 0.f);
    for (int ky = 0; ky < 5; ++ky)
      for (int kx = 0; kx < 5; ++kx) {
        float* dst = L.h_w.data() + (size_t)(ky * 5 + kx) * slab;
        for (int ci = 0; ci < cin; ++ci)
          for (int co = 0; co < cout; ++co) dst[(size_t)ci * L.Cin + co] = wref(ci, co, ky, kx);
      }
  }
  // batch-1 streaming layouts (kernels_b1.hip): [class][16-channel output slice][tap][32-channel step][lane][8], lane =
  // (n = l & 15, kg = l >> 4) holding reduction channels step*32 + kg*8 + 0..7 of output channel slice*16 + n
  if ((cin % 32) == 0 && (cout % 32) == 0 && (H % 4) == 0 && (W % 4) == 0) {
    op.h_b1_fwd.assign((size_t)25 * cin * cout, 0.f);
    size_t off = 0;
    for (int cls = 0; cls < 4; ++cls) {
      const int py = cls >> 1, px = cls & 1, nky = py ? 2 : 3, nkx = px ? 2 : 3;
      op.b1_cls_off[cls] = (long long)off;
      for (int sl = 0; sl < cout / 16; ++sl)
        for (int t = 0; t < nky * nkx; ++t) {
          const int ky = py + 2 * (t / nkx), kx = px + 2 * (t % nkx);
          for (int ks = 0; ks < cin / 32; ++ks, off += 512)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 8; ++j)
                op.h_b1_fwd[off + (size_t)l * 8 + j] = wref(ks * 32 + (l >> 4) * 8 + j, sl * 16 + (l & 15), ky, kx);
        }
    }
    op.h_b1_bwd.assign((size_t)25 * cin * cout, 0.f);
    off = 0;
    for (int sl = 0; sl < cin / 16; ++sl)
      for (int t = 0; t < 25; ++t)
        for (int ks = 0; ks < cout / 32; ++ks, off += 512)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j)
              op.h_b1_bwd[off + (size_t)l * 8 + j] = wref(sl * 16 + (l & 15), ks * 32 + (l >> 4) * 8 + j, t / 5, t % 5);
  }
  return 0;
}

// MDCL composite stencil (layers.py:207-258): tap list and, for every (3x3 branch, p, q), the tap it lands on.
// Branch 0 = the base 3x3 (dilation 1), then one branch per scale > 0 in config order; scale 0 = the 1x1 mean
// branch, which adds to the centre tap (tap 0).
struct MdcTable {
  std::vector<TgTap> taps;
  std::vector<int> dil;                 // dilation of each 3x3 branch
  std::vector<std::vector<int>> tapof;  // [branch][p*3+q] -> tap id
  bool has_1x1 = false;
};
MdcTable mdc_table(const ian_op_desc& d) {
  MdcTable T;
  T.dil.push_back(1);
  for (int i = 0; i < d.n_scales; ++i) {
    if (d.scales[i] == 0) T.has_1x1 = true;
    else T.dil.push_back(d.scales[i]);
  }
  std::map<std::pair<int, int>, int> index;
  auto tap_id = [&](int dy, int dx) {
    auto key = std::make_pair(dy, dx);
    auto it = index.find(key);
    if (it != index.end()) return it->second;
    const int id = (int)T.taps.size();
    index[key] = id;
    T.taps.push_back({dy, dx});
    return id;
  };
  tap_id(0, 0);
  for (int dl : T.dil) {
    std::vector<int> m(9);
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) m[p * 3 + q] = tap_id(dl * (p - 1), dl * (q - 1));
    T.tapof.push_back(m);
  }
  return T;
}

// MDC3: layers.py:207-258 collapsed into ONE composite sparse stencil (the idea sketched -- and broken --
// in layers.py:138-150 mdclW): offsets d*(p-1,q-1) for d in {1} + {s>0}, centre tap shared by all
// branches and by the 1x1 mean branch.  slab[offset][co][ci] = sum_b coeff_b[co] * W[co,ci,p,q].
int pack_mdc(ian_handle* h, OpPlan& op) {
  const int cin = op.d.cin, cout = op.d.cout, H = op.d.in_h, W = op.d.in_w;
  const HostTensor *Wt, *cb;
  int rc;
  if ((rc = need_param(h, op.name + "W", {cout, cin, 3, 3}, &Wt))) return rc;
  if ((rc = need_param(h, op.name + "_coeff_base", {cout}, &cb))) return rc;
  struct Branch {
    int d;
    const HostTensor* coeff;
  };
  std::vector<Branch> br;
  br.push_back({1, cb});
  const HostTensor* c1x1 = nullptr;
  for (int i = 0; i < op.d.n_scales; ++i) {
    const int s = op.d.scales[i];
    const HostTensor* c;
    if (s == 0) {
      if ((rc = need_param(h, op.name + "_coeff_1x1", {cout}, &c))) return rc;
      c1x1 = c;
    } else {
      if ((rc = need_param(h, op.name + "_coeff_" + std::to_string(s), {cout}, &c))) return rc;
      br.push_back({s, c});
    }
  }
  std::map<std::pair<int, int>, int> index;  // offset -> tap id
  std::vector<TgTap> taps;
  auto tap_id = [&](int dy, int dx) {
    auto key = std::make_pair(dy, dx);
    auto it = index.find(key);
    if (it != index.end()) return it->second;
    const int id = (int)taps.size();
    index[key] = id;
    taps.push_back({dy, dx});
    return id;
  };
  tap_id(0, 0);
  for (auto& b : br)
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) tap_id(b.d * (p - 1), b.d * (q - 1));
  const int nt = (int)taps.size();
  auto fill = [&](TgLayer& L, bool transpose) {
    L.valid = true;
    L.IH = H; L.IW = W; L.QH = H; L.QW = W; L.si = 1; L.by = 0; L.bx = 0; L.so = 1; L.OH = H; L.OW = W;
    const int kin = transpose ? cout : cin, kout = transpose ? cin : cout;
    L.Cin = round_up(kin, 32); L.cin_real = kin; L.Cout = kout; L.CoutPad = round_up(kout, 128);
    const size_t slab = (size_t)L.CoutPad * L.Cin;
    L.h_w.assign(slab * nt, 0.f);
    auto acc = [&](int t, int co, int ci, float v) {
      if (!transpose) L.h_w[t * slab + (size_t)co * L.Cin + ci] += v;
      else L.h_w[t * slab + (size_t)ci * L.Cin + co] += v;
    };
    for (auto& b : br)
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
          const int t = index[std::make_pair(b.d * (p - 1), b.d * (q - 1))];
          for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
              acc(t, co, ci, b.coeff->data[co] * Wt->data[(((size_t)co * cin + ci) * 3 + p) * 3 + q]);
        }
    if (c1x1) {
      for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
          float m = 0.f;
          for (int k = 0; k < 9; ++k) m += Wt->data[((size_t)co * cin + ci) * 9 + k];
          acc(0, co, ci, c1x1->data[co] * (m / 9.f));
        }
    }
    std::vector<TgTap> tt = taps;
    if (transpose)
      for (auto& t : tt) { t.dy = -t.dy; t.dx = -t.dx; }
    add_class(L, 0, 0, tt, 0);
  };
  fill(op.fwd, false);
  fill(op.bwd, true);
  return 0;
}

// DENSE: x.W (+b); W (in,out).  forward slab [CoutPad][CinPad] = W^T with the (C,H,W)->(H,W,C) permutations
// of App. B.6 baked in; backward-data slab [in][out].
int pack_dense(ian_handle* h, OpPlan& op, std::vector<int>& out_perm, bool& has_out_perm) {
  const int fin = op.d.cin, fout = op.d.cout;
  const HostTensor* Wt;
  int rc = need_param(h, op.name + ".W", {fin, fout}, &Wt);
  if (rc) return rc;
  std::vector<int> in_perm;
  const bool has_in = op.d.flat_c > 0;
  if (has_in) in_perm = chw_to_hwc_perm(op.d.flat_c, op.d.flat_h, op.d.flat_w);
  has_out_perm = op.d.unflat_c > 0;
  if (has_out_perm) out_perm = chw_to_hwc_perm(op.d.unflat_c, op.d.unflat_h, op.d.unflat_w);
  auto setup = [&](TgLayer& L, int kin, int kout) {
    L.valid = true;
    L.Cin = round_up(kin, 32); L.cin_real = kin; L.Cout = kout; L.CoutPad = round_up(kout, 128);
    add_class(L, 0, 0, {{0, 0}}, 0);
    L.h_w.assign((size_t)L.CoutPad * L.Cin, 0.f);
  };
  setup(op.fwd, fin, fout);
  setup(op.bwd, fout, fin);
  for (int i = 0; i < fin; ++i) {
    const int ii = has_in ? in_perm[i] : i;
    for (int j = 0; j < fout; ++j) {
      const int jj = has_out_perm ? out_perm[j] : j;
      const float v = Wt->data[(size_t)i * fout + j];
      op.fwd.h_w[(size_t)jj * op.fwd.Cin + ii] = v;
      op.bwd.h_w[(size_t)ii * op.bwd.Cin + jj] = v;
    }
  }
  return 0;
}

int pack_made(ian_handle* h, OpPlan& op) {
  const int d = op.d.cin;
  if (h->made_n != d) return fail(h, -3, "MADE masks not set (ian_set_made_masks) or wrong size");
  static const char* which[2] = {"_mu", "_ls"};
  static const char* lay[3] = {"_input", "_output_W", "_output_D"};
  const int mask_of[3] = {0, 1, 2};
  std::vector<float> w((size_t)6 * d * d), b((size_t)6 * d);
  for (int m = 0; m < 2; ++m)
    for (int l = 0; l < 3; ++l) {
      const std::string base = op.name + which[m] + lay[l];
      const HostTensor *Wt, *bt;
      int rc;
      if ((rc = need_param(h, base + ".W", {d, d}, &Wt))) return rc;
      if ((rc = need_param(h, base + ".b", {d}, &bt))) return rc;
      const std::vector<float>& mask = h->made_masks[mask_of[l]];
      float* wd = w.data() + (size_t)(m * 3 + l) * d * d;
      for (int i = 0; i < d * d; ++i) wd[i] = Wt->data[i] * mask[i];  // layers.py:671,703: W * weights_mask
      for (int i = 0; i < d; ++i) b[(size_t)(m * 3 + l) * d + i] = bt->data[i];
    }
  int rc;
  if ((rc = upload(h, w, &op.d_made_w))) return rc;
  return upload(h, b, &op.d_made_b);
}

int upload_layer(ian_handle* h, TgLayer& L) {
  if (!L.valid) return 0;
  if (ilog2_exact(L.QW) < 0 || ilog2_exact(L.QH * L.QW) < 0) return fail(h, -4, "tapgemm needs power-of-two output grids");
  int rc;
  L.w_floats = L.h_w.size();
  if (L.w_floats * 4 > 0xFFFFFFF0ull) return fail(h, -4, "packed weights exceed the 4 GiB buffer-descriptor range");
  if ((rc = upload(h, L.h_w, &L.d_w))) return rc;
  if ((rc = upload(h, L.classes, &L.d_classes))) return rc;
  if ((rc = upload(h, L.taps, &L.d_taps))) return rc;
  std::vector<float>().swap(L.h_w);
  return 0;
}

// ----- scheduling: tiles, split-K, heavy-first + XCD-aware item order ---------------------------------
static int count_tiles(const TgLayer& L, int M, int cfg) {
  const TgShape sh = tg_shape(cfg);
  return ((M + sh.bm - 1) / sh.bm) * ((L.Cout + sh.bn - 1) / sh.bn) * (int)L.classes.size();
}

int pick_config(const ian_handle* h, const TgLayer& L, int M) {
  if (h->opt.tg_cfg >= 0 && h->opt.tg_cfg < TG_NCONFIG) {
    const TgShape s = tg_shape(h->opt.tg_cfg);
    if (L.CoutPad % s.bn == 0) return h->opt.tg_cfg;
  }
  if (L.Cout <= 32) return TG_128x32;
  if (M <= 32) return TG_32x128;
  if (M <= 64) return TG_64x64;
  // Large output maps: a smaller tile that fills the chip WITHOUT split-K beats 128x128 + slabs (the slab
  // write + reduce pass moves the whole output several times).  Small-M / huge-K layers keep 128x128 + split-K.
  if (h->opt.tg_prefer_nosplit) {
    const int order[3] = {TG_128x128, TG_128x64, TG_64x64};
    for (int c : order)
      if (count_tiles(L, M, c) >= h->opt.tg_no_split_items) return c;
    if ((long long)M * L.Cout >= (long long)h->opt.tg_nosplit_min_out) return TG_64x64;
  }
  return TG_128x128;
}

void build_schedule(const ian_handle* h, const TgLayer& L, int nimg, Schedule& S, const TgChoice& ch) {
  const int M = nimg * L.QH * L.QW;
  S.cfg = (ch.cfg >= 0 && ch.cfg < TG_NCONFIG && L.CoutPad % tg_shape(ch.cfg).bn == 0) ? ch.cfg : pick_config(h, L, M);
  S.variant = ch.variant;
  const TgShape sh = tg_shape(S.cfg);
  const int tiles_m = (M + sh.bm - 1) / sh.bm;
  const int tiles_n = (L.Cout + sh.bn - 1) / sh.bn;
  const int kpt = L.Cin / 32;
  const int ncls = (int)L.classes.size();
  long long total_steps = 0;
  int total_tiles = 0;
  for (auto& c : L.classes) {
    total_steps += (long long)tiles_m * tiles_n * c.ntaps * kpt;
    total_tiles += tiles_m * tiles_n;
  }
  int steps_per_item = 1 << 30;
  bool split = false;
  if (ch.max_steps > 0) {
    steps_per_item = ch.max_steps;
  } else if (ch.max_steps < 0 && h->opt.tg_split && total_tiles < h->opt.tg_no_split_items) {
    long long spi = (total_steps + h->opt.tg_target_items - 1) / h->opt.tg_target_items;
    steps_per_item = (int)std::max<long long>(spi, h->opt.tg_min_steps);
  }
  for (auto& c : L.classes)
    if (c.ntaps * kpt > steps_per_item) split = true;
  struct Group {
    std::vector<TgItem> items;
    int weight;
  };
  std::vector<Group> groups;
  int gm = std::max(1, h->opt.tg_xcd_group), gn = std::max(1, h->opt.tg_xcd_group);
  S.h_tiles.clear();
  S.max_nsplit = 1;
  size_t slab_next = 0;
  // tile -> slab bookkeeping
  std::vector<std::vector<int>> nsplit_c(ncls);
  for (int c = 0; c < ncls; ++c) {
    const int ksteps = L.classes[c].ntaps * kpt;
    int ns = 1;
    if (split) ns = std::max(1, (ksteps + steps_per_item - 1) / steps_per_item);
    const int per = (ksteps + ns - 1) / ns;
    ns = (ksteps + per - 1) / per;
    if (split) S.max_nsplit = std::max(S.max_nsplit, ns);
    // slab indices: tile-major so that the reduce pass reads contiguous slabs
    std::vector<int> slab0((size_t)tiles_m * tiles_n, -1), tile_id((size_t)tiles_m * tiles_n, -1);
    if (split) {
      for (int mt = 0; mt < tiles_m; ++mt)
        for (int nt = 0; nt < tiles_n; ++nt) {
          slab0[(size_t)mt * tiles_n + nt] = (int)slab_next;
          tile_id[(size_t)mt * tiles_n + nt] = (int)S.h_tiles.size();
          TgTile t{c, mt * sh.bm, nt * sh.bn, (int)slab_next, ns, 0, 0, 0};
          S.h_tiles.push_back(t);
          slab_next += ns;
        }
    }
    for (int s = 0; s < ns; ++s) {
      const int k0 = s * per, k1 = std::min(ksteps, (s + 1) * per);
      for (int nb = 0; nb < tiles_n; nb += gn)
        for (int mb = 0; mb < tiles_m; mb += gm) {
          Group g;
          g.weight = k1 - k0;
          for (int nt = nb; nt < std::min(tiles_n, nb + gn); ++nt)
            for (int mt = mb; mt < std::min(tiles_m, mb + gm); ++mt) {
              TgItem it{c, mt * sh.bm, nt * sh.bn, k0, k1, split ? slab0[(size_t)mt * tiles_n + nt] + s : -1,
                        tile_id[(size_t)mt * tiles_n + nt], 0};
              g.items.push_back(it);
            }
          groups.push_back(std::move(g));
        }
    }
  }
  std::stable_sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) { return a.weight > b.weight; });
  // deal supergroups to the 8 XCDs (block b runs on XCD b%8): always to the least-loaded list
  std::vector<std::vector<TgItem>> lists(8);
  std::vector<long long> load(8, 0);
  for (auto& g : groups) {
    int best = 0;
    for (int x = 1; x < 8; ++x)
      if (load[x] < load[best]) best = x;
    for (auto& it : g.items) lists[best].push_back(it);
    load[best] += (long long)g.weight * g.items.size();
  }
  size_t longest = 0;
  for (auto& l : lists) longest = std::max(longest, l.size());
  S.h_items.clear();
  const TgItem empty{0, 0, 0, 0, 0, -1, 0, 0};
  for (size_t k = 0; k < longest; ++k)
    for (int x = 0; x < 8; ++x) S.h_items.push_back(k < lists[x].size() ? lists[x][k] : empty);
  while (!S.h_items.empty() && S.h_items.back().ks0 >= S.h_items.back().ks1) S.h_items.pop_back();
  S.nitems = (int)S.h_items.size();
  S.ntiles = split ? (int)S.h_tiles.size() : 0;
  S.slab_tiles = slab_next;
}

int get_schedule(ian_handle* h, TgLayer& L, int nimg, Schedule** out) {
  auto it = L.sched.find(nimg);
  if (it == L.sched.end()) {
    Schedule S;
    auto chit = L.choice.find(nimg);
    build_schedule(h, L, nimg, S, chit == L.choice.end() ? TgChoice() : chit->second);
    int rc;
    if ((rc = upload(h, S.h_items, &S.d_items))) return rc;
    if ((rc = upload(h, S.h_tiles, &S.d_tiles))) return rc;
    if (S.ntiles > 0) {
      HIPCHK(h, hipMalloc((void**)&S.d_counters, S.ntiles * sizeof(int)));
      HIPCHK(h, hipMemset(S.d_counters, 0, S.ntiles * sizeof(int)));
    }
    const TgShape sh = tg_shape(S.cfg);
    const size_t need = S.slab_tiles * sh.bm * sh.bn;
    if (need > h->slab_cap) {
      if (h->d_slab) HIPCHK(h, hipFree(h->d_slab));
      HIPCHK(h, hipMalloc((void**)&h->d_slab, need * sizeof(float)));
      h->slab_cap = need;
      ++h->alloc_epoch;
    }
    it = L.sched.emplace(nimg, std::move(S)).first;
  }
  *out = &it->second;
  return 0;
}

void free_schedule_for(TgLayer& L, int nimg) {
  auto it = L.sched.find(nimg);
  if (it == L.sched.end()) return;
  if (it->second.d_items) (void)hipFree(it->second.d_items);
  if (it->second.d_tiles) (void)hipFree(it->second.d_tiles);
  if (it->second.d_counters) (void)hipFree(it->second.d_counters);
  L.sched.erase(it);
}

void free_schedules(TgLayer& L) {
  for (auto& kv : L.sched) {
    if (kv.second.d_items) (void)hipFree(kv.second.d_items);
    if (kv.second.d_tiles) (void)hipFree(kv.second.d_tiles);
    if (kv.second.d_counters) (void)hipFree(kv.second.d_counters);
  }
  L.sched.clear();
}

int ensure_slot(ian_handle* h, int slot, int n, bool grad = false) {
  Slot& s = h->slots[slot];
  const size_t need = s.per_image() * (size_t)n;
  float*& ptr = grad ? s.g : s.d;
  size_t& cap = grad ? s.gcap : s.cap;
  if (need > cap) {
    if (ptr) HIPCHK(h, hipFree(ptr));
    HIPCHK(h, hipMalloc((void**)&ptr, need * sizeof(float)));
    HIPCHK(h, hipMemset(ptr, 0, need * sizeof(float)));  // channel padding must stay zero
    cap = need;
    ++h->alloc_epoch;
  }
  return 0;
}

bool is_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

int run_tapgemm(ian_handle* h, TgLayer& L, int nimg, const float* x, float* y, int y_stride, const TgEpilogue& epi,
                hipStream_t st) {
  Schedule* S;
  int rc = get_schedule(h, L, nimg, &S);
  if (rc) return rc;
  TgParams p;
  p.x = x; p.w = L.d_w; p.y = y; p.slab = h->d_slab;
  p.items = S->d_items; p.classes = L.d_classes; p.taps = L.d_taps;
  p.epi = epi;
  p.M = nimg * L.QH * L.QW;
  p.IH = L.IH; p.IW = L.IW; p.Cin = L.Cin;
  p.qw_shift = ilog2_exact(L.QW); p.qhw_shift = ilog2_exact(L.QH * L.QW);
  p.si = L.si; p.by = L.by; p.bx = L.bx; p.so = L.so;
  p.OH = L.OH; p.OW = L.OW; p.Cout = L.Cout; p.y_stride = y_stride; p.CoutPad = L.CoutPad;
  const size_t xb = (size_t)nimg * L.IH * L.IW * L.Cin * sizeof(float);
  if (xb > 0xFFFFFFF0ull) return fail(h, -7, "batch %d makes a %zu-byte activation: above the 4 GiB buffer-descriptor range, split the batch", nimg, xb);
  p.x_bytes = (unsigned)xb;
  p.w_bytes = (unsigned)(L.w_floats * sizeof(float));
  p.variant = S->variant >= 0 ? S->variant : h->opt.tg_variant;
  const bool fused = S->ntiles > 0 && p.M <= h->opt.tg_fused_reduce_max_m;
  p.tiles = S->d_tiles;
  p.counters = fused ? S->d_counters : nullptr;
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  if (h->prof) {
    if (h->ev_used == h->ev_pool.size()) {
      hipEvent_t a, b;
      HIPCHK(h, hipEventCreate(&a));
      HIPCHK(h, hipEventCreate(&b));
      h->ev_pool.push_back({a, b});
    }
    ev = &h->ev_pool[h->ev_used++];
    HIPCHK(h, hipEventRecord(ev->first, st));
  }
  HIPCHK(h, launch_tapgemm(S->cfg, p, S->nitems, st));
  if (S->ntiles > 0 && !fused) {
    TgReduceParams r;
    r.slab = h->d_slab; r.y = y; r.tiles = S->d_tiles; r.classes = L.d_classes; r.epi = epi;
    r.M = p.M; r.qw_shift = p.qw_shift; r.qhw_shift = p.qhw_shift; r.so = L.so; r.OH = L.OH; r.OW = L.OW;
    r.Cout = L.Cout; r.y_stride = y_stride;
    HIPCHK(h, launch_tapgemm_reduce(S->cfg, r, S->ntiles, (S->max_nsplit >= 8 && h->opt.tg_reduce_kp > 1) ? 4 : 1, st));
  }
  if (ev) {
    HIPCHK(h, hipEventRecord(ev->second, st));
    h->prof_flops += 2.0 * L.macs_per_image() * nimg;
    h->prof_launches += 1;
  }
  return 0;
}

// thin MDCL: <= 4 real input channels (forward of G_b/B_b; backward-data form of every 2-filter head layer)
bool mdc_thin_eligible(const ian_handle* h, const TgLayer& L) {
  if (!h->opt.mdc_head || !L.valid || L.cin_real > 4 || L.classes.size() != 1 || L.taps.size() > 36) return false;
  if (L.si != 1 || L.so != 1 || L.by != 0 || L.bx != 0 || L.QH != L.IH || L.QW != L.IW) return false;
  if (L.Cout > 4) return false;  // wide outputs (backward-data of the 2-filter heads) measured 10x slower here than on tapgemm
  for (auto& t : L.taps)
    if (t.dy < -64 || t.dy > 64 || t.dx < -64 || t.dx > 64) return false;
  return true;
}

void mdc_thin_fill(MdcThinArgs& a, const TgLayer& L, const float* x, int xs, float* y, int ys, int n, const ian_handle* h) {
  memset(&a, 0, sizeof a);
  a.no_tile = h->opt.mdc_thin_tile ? 0 : 1;
  a.x = x; a.w = L.d_w; a.y = y; a.n = n; a.H = L.IH; a.W = L.IW; a.xs = xs; a.ys = ys; a.Cout = L.Cout;
  a.CoutPad = L.CoutPad; a.CinPad = L.Cin; a.ntaps = (int)L.taps.size();
  for (int t = 0; t < a.ntaps; ++t) { a.dy[t] = (signed char)L.taps[t].dy; a.dx[t] = (signed char)L.taps[t].dx; }
}

bool mdc_head_eligible(const ian_handle* h, const OpPlan& op) {
  const TgLayer& L = op.fwd;
  if (!h->opt.mdc_head || L.Cout > 4 || (L.Cin != 128 && L.Cin != 64) || L.cin_real != L.Cin) return false;
  if ((L.IH % 4) || (L.IW % 16) || L.taps.size() > 48 || L.classes.size() != 1) return false;
  for (auto& t : L.taps)
    if (t.dy < -4 || t.dy > 4 || t.dx < -4 || t.dx > 4) return false;
  return true;
}

// Fill the per-filter slots of MdcHeadArgs for one layer; weights/scale/shift are read back from the host copies.
int mdc_head_add(ian_handle* h, MdcHeadArgs& a, int& nco, const OpPlan& op, float* y, int ys, const float* res) {
  const TgLayer& L = op.fwd;
  for (int co = 0; co < L.Cout; ++co) {
    if (nco >= MH_MAXCO) return fail(h, -4, "mdc head: too many filters");
    a.w[nco] = L.d_w + (size_t)co * L.Cin;
    a.res[nco] = res;
    a.y[nco] = y;
    a.ys[nco] = ys;
    a.yc[nco] = co;
    a.act[nco] = op.d.act;
    a.scale[nco] = op.h_scale.empty() ? 1.f : op.h_scale[co];
    a.shift[nco] = op.h_shift.empty() ? 0.f : op.h_shift[co];
    ++nco;
  }
  return 0;
}

// Inference: few-filter MDCL layers on the VALU kernel.  Sibling layers that read the same map with the same taps and
// have no residual input are computed by ONE launch when the first of them is reached (their inputs are ready: same
// source); the others are skipped when their turn comes (OpPlan::head_done_for marks the batch they were computed for).
int run_mdc_head_group(ian_handle* h, OpPlan& op, int n, hipStream_t st) {
  if (op.head_done_serial == h->run_serial) return 0;  // already produced by its group leader in this call
  const TgLayer& L = op.fwd;
  MdcHeadArgs a;
  memset(&a, 0, sizeof a);
  Slot& src = h->slots[op.d.src];
  a.x = src.d; a.H = L.IH; a.W = L.IW; a.xs = src.cs; a.ntaps = (int)L.taps.size();
  a.w_tap_stride = (long long)L.CoutPad * L.Cin;
  for (int t = 0; t < a.ntaps; ++t) { a.dy[t] = (signed char)L.taps[t].dy; a.dx[t] = (signed char)L.taps[t].dx; }
  int nco = 0, rc;
  Slot& dst = h->slots[op.d.dst];
  const float* res = op.d.src2 >= 0 ? h->slots[op.d.src2].d : nullptr;
  if ((rc = mdc_head_add(h, a, nco, op, dst.d, dst.cs, res))) return rc;
  if (h->opt.mdc_head >= 2 && op.d.src2 < 0) {
    for (auto& o : h->ops) {
      if (&o == &op || o.d.kind != IAN_OP_MDC3 || o.d.segment != op.d.segment || o.d.src != op.d.src || o.d.src2 >= 0) continue;
      if (o.head_done_serial == h->run_serial || !mdc_head_eligible(h, o)) continue;
      const TgLayer& M = o.fwd;
      if (M.taps.size() != L.taps.size() || M.Cin != L.Cin || M.CoutPad != L.CoutPad || nco + M.Cout > MH_MAXCO) continue;
      bool same = true;
      for (size_t t = 0; t < L.taps.size(); ++t) same = same && M.taps[t].dy == L.taps[t].dy && M.taps[t].dx == L.taps[t].dx;
      if (!same) continue;
      if ((rc = ensure_slot(h, o.d.dst, n))) return rc;
      Slot& od = h->slots[o.d.dst];
      if ((rc = mdc_head_add(h, a, nco, o, od.d, od.cs, nullptr))) return rc;
      o.head_done_serial = h->run_serial;
    }
  }
  HIPCHK(h, launch_mdc_head(a, n, L.Cin, nco, st));
  return 0;
}

// ----- fused RGB-Beta head (kernels_head.hip) --------------------------------------------------------------------
bool same_taps(const TgLayer& A, const TgLayer& B) {
  if (A.taps.size() != B.taps.size() || A.classes.size() != 1 || B.classes.size() != 1) return false;
  for (size_t t = 0; t < A.taps.size(); ++t)
    if (A.taps[t].dy != B.taps[t].dy || A.taps[t].dx != B.taps[t].dx) return false;
  return true;
}

int find_head_plan(ian_handle* h) {
  HeadPlan& P = h->head;
  P.searched = true;
  const int nops = (int)h->ops.size();
  std::vector<int> prod(h->slots.size(), -1);
  for (int i = 0; i < nops; ++i) prod[h->ops[i].d.dst] = i;
  auto mdc2 = [&](int i) { return i >= 0 && h->ops[i].d.kind == IAN_OP_MDC3 && h->ops[i].d.cout == 2 && h->ops[i].d.segment == IAN_SEG_DEC; };
  for (int ib = 0; ib < nops; ++ib) {
    const ian_op_desc& b = h->ops[ib].d;
    if (b.kind != IAN_OP_BETA || b.dst != h->desc.out_slot || b.src < 0 || b.src2 < 0 || b.src3 < 0) continue;
    const int iR = prod[b.src], iGb = prod[b.src2], iBb = prod[b.src3];
    if (!mdc2(iR) || !mdc2(iGb) || !mdc2(iBb)) continue;
    const ian_op_desc &R = h->ops[iR].d, &Gb = h->ops[iGb].d, &Bb = h->ops[iBb].d;
    if (R.src2 >= 0 || Gb.src != R.dst || Gb.src2 < 0 || Bb.src2 < 0 || Gb.cin != 2 || Bb.cin != 4) continue;
    const int iGa = prod[Gb.src2], iBa = prod[Bb.src2], iCat = prod[Bb.src];
    if (!mdc2(iGa) || !mdc2(iBa) || iCat < 0) continue;
    const ian_op_desc &Ga = h->ops[iGa].d, &Ba = h->ops[iBa].d, &Cat = h->ops[iCat].d;
    if (Cat.kind != IAN_OP_CONCAT || Cat.src != R.dst || Cat.src2 != Gb.dst) continue;
    if (Ga.src != R.src || Ba.src != R.src || Ga.src2 >= 0 || Ba.src2 >= 0) continue;
    if (R.cin != 128 || R.in_w != 64 || h->slots[R.src].cs != 128) continue;
    const TgLayer& LR = h->ops[iR].fwd;
    bool ok = same_taps(LR, h->ops[iGa].fwd) && same_taps(LR, h->ops[iBa].fwd) && same_taps(LR, h->ops[iGb].fwd) &&
              same_taps(LR, h->ops[iBb].fwd) && LR.taps.size() <= 37;
    int halo = 0;
    std::vector<int> per_dy(9, 0);
    for (auto& t : LR.taps) {
      if (t.dy < -4 || t.dy > 4 || t.dx < -64 || t.dx > 64) ok = false;
      else if (++per_dy[t.dy + 4] > 12) ok = false;
      halo = std::max(halo, std::abs(t.dy));
    }
    // each of the three slots must feed nothing but the head (they are never materialised by the fused launch)
    for (int i = 0; i < nops && ok; ++i) {
      const ian_op_desc& d = h->ops[i].d;
      for (int sl : {Ga.dst, Ba.dst, Cat.dst})
        if ((d.src == sl || d.src2 == sl || d.src3 == sl) && i != iGb && i != iBb) ok = false;
    }
    if (!ok) continue;
    P.opR = iR; P.opGa = iGa; P.opGb = iGb; P.opBa = iBa; P.opBb = iBb; P.opCat = iCat; P.opBeta = ib; P.halo = halo;
    P.first = std::min({iR, iGa, iBa});
    // tables: taps grouped by dy (the shift-add of head6_kernel) and the six filters' epilogues
    std::vector<int> itab(44, -1);   // head6_kernel's static slot layout
    std::vector<int> used(9, 0);
    for (size_t t = 0; t < LR.taps.size() && ok; ++t) {
      const int dy = LR.taps[t].dy, g = dy == 0 ? 0 : (dy < 0 ? dy + 5 : dy + 4);
      const int cap = g == 0 ? 12 : 4, s0 = g == 0 ? 0 : 12 + (g - 1) * 4;
      if (used[g] >= cap || LR.taps[t].dx < -63 || LR.taps[t].dx > 63) { ok = false; break; }
      itab[s0 + used[g]++] = (int)t | ((LR.taps[t].dx + 64) << 8);
    }
    if (!ok) continue;
    std::vector<float> ftab(32, 0.f);
    const int six[3] = {iR, iGa, iBa};
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 2; ++c) {
        const OpPlan& o = h->ops[six[k]];
        ftab[2 * k + c] = o.h_scale.empty() ? 1.f : o.h_scale[c];
        ftab[8 + 2 * k + c] = o.h_shift.empty() ? 0.f : o.h_shift[c];
        ftab[16 + 2 * k + c] = (float)o.d.act;
      }
    int rc;
    if ((rc = upload(h, itab, &P.d_itab))) return rc;
    if ((rc = upload(h, ftab, &P.d_ftab))) return rc;
    P.valid = true;
    return 0;
  }
  return 0;
}

bool head_fused_active(const ian_handle* h, int n) {
  return h->head.valid && h->opt.head_fused && h->opt.mdc_head && n >= h->opt.head_fused_min_n;
}

int run_head_fused(ian_handle* h, int n, hipStream_t st) {
  HeadPlan& P = h->head;
  const OpPlan &R = h->ops[P.opR], &Ga = h->ops[P.opGa], &Gb = h->ops[P.opGb], &Ba = h->ops[P.opBa], &Bb = h->ops[P.opBb];
  const Slot& src = h->slots[R.d.src];
  const int H = R.d.in_h, W = R.d.in_w;
  const size_t need = (size_t)n * H * W * 8;
  if (need > P.comp_cap) {
    if (P.d_comp) HIPCHK(h, hipFree(P.d_comp));
    HIPCHK(h, hipMalloc((void**)&P.d_comp, need * sizeof(float)));
    P.comp_cap = need;
  }
  int rc;
  if ((rc = ensure_slot(h, h->desc.out_slot, n))) return rc;
  HeadFusedArgs a;
  memset(&a, 0, sizeof a);
  a.x = src.d; a.w0 = R.fwd.d_w; a.w1 = Ga.fwd.d_w; a.w2 = Ba.fwd.d_w; a.out = P.d_comp; a.itab = P.d_itab; a.ftab = P.d_ftab;
  a.H = H; a.W = W; a.xs = src.cs; a.ntaps = (int)R.fwd.taps.size(); a.halo = P.halo;
  a.w_tap_stride = (long long)R.fwd.CoutPad * R.fwd.Cin;
  // bands: whole images per workgroup when the batch alone fills the chip (no halo rows recomputed); smaller batches
  // are cut into row bands so that at least ~256 workgroups exist (each band re-reads 2*halo rows)
  int bands = 1;
  while (n * bands < 256 && bands < 8 && (H % (bands * 2)) == 0 && H / (bands * 2) >= 2 * P.halo) bands *= 2;
  a.bands = bands;
  HIPCHK(h, launch_head6(a, n, st));
  HeadTailArgs t;
  memset(&t, 0, sizeof t);
  t.comp = P.d_comp; t.w_gb = Gb.fwd.d_w; t.w_bb = Bb.fwd.d_w; t.out = h->slots[h->desc.out_slot].d;
  t.gb_tap_stride = (long long)Gb.fwd.CoutPad * Gb.fwd.Cin; t.bb_tap_stride = (long long)Bb.fwd.CoutPad * Bb.fwd.Cin;
  t.gb_cin = Gb.fwd.Cin; t.bb_cin = Bb.fwd.Cin;
  for (int c = 0; c < 2; ++c) {
    t.scale_g[c] = Gb.h_scale.empty() ? 1.f : Gb.h_scale[c]; t.shift_g[c] = Gb.h_shift.empty() ? 0.f : Gb.h_shift[c];
    t.scale_b[c] = Bb.h_scale.empty() ? 1.f : Bb.h_scale[c]; t.shift_b[c] = Bb.h_shift.empty() ? 0.f : Bb.h_shift[c];
  }
  t.act_g = Gb.d.act; t.act_b = Bb.d.act; t.H = H; t.W = W; t.ntaps = a.ntaps;
  for (int k = 0; k < a.ntaps; ++k) { t.dy[k] = (signed char)R.fwd.taps[k].dy; t.dx[k] = (signed char)R.fwd.taps[k].dx; }
  HIPCHK(h, launch_head_tail(t, n, st));
  return 0;
}

TgEpilogue fwd_epi(const OpPlan& op, const float* res) {
  TgEpilogue e;
  e.scale = op.d_scale; e.shift = op.d_shift; e.res = res; e.yfwd = nullptr; e.act = op.d.act; e.mode = TG_EPI_FWD;
  e.scale_period = 0;
  return e;
}

// batch-1 transposed conv / its backward-data as one whole-contraction streaming launch (kernels_b1.hip).
// mode 0: x = the layer input (H x W x cin), y = its output;  mode 1: x = dL/d(pre-epilogue output) (2H x 2W x cout), y = dL/d(input)
bool b1_fill(const ian_handle* h, const OpPlan& op, int mode, const Slot& xin, const Slot& yout, B1Params& p) {
  if (!h->opt.b1_conv || op.d.kind != IAN_OP_DECONV5S2 || op.edge) return false;
  const float* w = mode == 0 ? op.d_b1_fwd : op.d_b1_bwd;
  if (!w) return false;
  const int H = op.d.in_h, W = op.d.in_w, cin = op.d.cin, cout = op.d.cout;
  p = B1Params();
  p.w = w;
  if (mode == 0) {
    p.IH = H; p.IW = W; p.Cr = cin; p.OH = 2 * H; p.OW = 2 * W;
    p.nslices = cout / 16;
    for (int c = 0; c < 4; ++c) p.cls_off[c] = op.b1_cls_off[c];
  } else {
    p.IH = 2 * H; p.IW = 2 * W; p.Cr = cout; p.OH = H; p.OW = W;
    p.nslices = cin / 16;
  }
  p.xs = xin.cs; p.ys = yout.cs;
  if (p.xs < p.Cr || (p.xs & 3)) return false;
  p.tiles_x = W / 4;
  p.ntiles = (H / 4) * (W / 4);
  const size_t xb = (size_t)p.IH * p.IW * p.xs * sizeof(float);
  if (xb > 0xFFFFFFC0ull) return false;
  p.x_bytes = (unsigned)xb;
  p.kshift = ilog2_exact(p.Cr / 32);
  return p.kshift >= 0;
}

// ----- forward executor ---------------------------------------------------------------------------------
int run_op_fwd(ian_handle* h, OpPlan& op, int n, hipStream_t st) {
  Slot& src = h->slots[op.d.src];
  Slot& dst = h->slots[op.d.dst];
  int rc;
  if ((rc = ensure_slot(h, op.d.dst, n))) return rc;
  const float* res = nullptr;
  switch (op.d.kind) {
    case IAN_OP_CONV5S2:
      if (op.edge) {
        HIPCHK(h, launch_conv1_nchw(src.d, op.d_edge_w, op.d_scale, op.d_shift, dst.d, n, op.d.in_h, op.d.in_w,
                                    op.d.cout, op.d.act, st));
        return 0;
      }
      return run_tapgemm(h, op.fwd, n, src.d, dst.d, dst.cs, fwd_epi(op, nullptr), st);
    case IAN_OP_DECONV5S2:
      if (op.edge && h->opt.dec_out_mfma && n >= 4 && op.d.cin == 128 && src.cs == 128 && op.d.in_w == 32 && (op.d.in_h % 2) == 0 &&
          (op.d.cout == 3 || op.d.cout == 4)) {
        // image-producing deconv on the matrix cores (kernels_head.hip: contract first, scatter the 25 taps later)
        DeconvSmallArgs a;
        a.x = src.d; a.w = op.d_edge_w; a.scale = op.d_scale; a.shift = op.d_shift; a.y = dst.d;
        a.H = op.d.in_h; a.W = op.d.in_w; a.xs = src.cs; a.act = op.d.act;
        int bands = 1;
        while (n * bands < h->opt.dec_out_wgs && bands < 8 && (op.d.in_h % (4 * bands)) == 0) bands *= 2;   // one halo row per side of a band
        a.bands = bands;
        HIPCHK(h, launch_deconv_small(a, n, op.d.cout, st));
        return 0;
      }
      if (op.edge && h->opt.dec_out_px && n < 4 && (src.cs == 64 || src.cs == 128 || src.cs == 256)) {
        HIPCHK(h, launch_deconv_out_px(src.d, op.d_edge_w, op.d_scale, op.d_shift, dst.d, n, op.d.in_h, op.d.in_w, src.cs,
                                       op.d.cout, op.d.act, st));
        return 0;
      }
      if (op.edge) {
        HIPCHK(h, launch_deconv_out_nchw(src.d, op.d_edge_w, op.d_scale, op.d_shift, dst.d, n, op.d.in_h, op.d.in_w,
                                         src.cs, op.d.cout, op.d.act, st));
        return 0;
      }
      if (n == 1) {
        B1Params bp;
        if (b1_fill(h, op, 0, src, dst, bp)) {
          bp.x = src.d; bp.y = dst.d; bp.scale = op.d_scale; bp.shift = op.d_shift; bp.act = op.d.act; bp.bwd = 0;
          HIPCHK(h, launch_b1conv(bp, 0, st));
          return 0;
        }
      }
      return run_tapgemm(h, op.fwd, n, src.d, dst.d, dst.cs, fwd_epi(op, nullptr), st);
    case IAN_OP_MDC3:
      if (op.d.src2 >= 0) res = h->slots[op.d.src2].d;
      if (mdc_head_eligible(h, op)) return run_mdc_head_group(h, op, n, st);
      if (mdc_thin_eligible(h, op.fwd)) {
        MdcThinArgs a;
        mdc_thin_fill(a, op.fwd, src.d, src.cs, dst.d, dst.cs, n, h);
        a.res = res; a.scale = op.d_scale; a.shift = op.d_shift; a.act = op.d.act;
        HIPCHK(h, launch_mdc_thin(a, st));
        return 0;
      }
      return run_tapgemm(h, op.fwd, n, src.d, dst.d, dst.cs, fwd_epi(op, res), st);
    case IAN_OP_DENSE:
      if (n == 1 && h->opt.dense_gemv && op.d.flat_c <= 0 && op.fwd.Cin <= 256 && (op.fwd.Cin & 31) == 0 && op.fwd.Cout >= 1024) {
        // the latent's own layer at batch 1 (interactive decoder): one short-vector GEMV launch, no split-K pass
        HIPCHK(h, launch_dense_fwd_gemv(src.d, op.fwd.d_w, op.fwd.Cin, op.fwd.Cout, op.d_scale, op.d_shift, op.d.act, dst.d, st));
        return 0;
      }
      return run_tapgemm(h, op.fwd, n, src.d, dst.d, (int)dst.per_image(), fwd_epi(op, nullptr), st);
    case IAN_OP_AFFINE:
      HIPCHK(h, launch_affine(src.d, dst.d, op.d_scale, op.d_shift, (long long)n * src.h * src.w, src.c, src.cs,
                              op.d.act, st));
      return 0;
    case IAN_OP_MADE_IAF:
      HIPCHK(h, launch_made_iaf(src.d, dst.d, op.d_made_w, op.d_made_b, n, op.d.cin, src.cs, st));
      return 0;
    case IAN_OP_BETA:
      HIPCHK(h, launch_beta(src.d, h->slots[op.d.src2].d, h->slots[op.d.src3].d, dst.d, n, src.h * src.w, src.cs, st));
      return 0;
    case IAN_OP_CONCAT: {
      Slot& b = h->slots[op.d.src2];
      HIPCHK(h, launch_concat2(src.d, src.c, src.cs, b.d, b.c, b.cs, dst.d, dst.cs, (long long)n * src.h * src.w, st));
      return 0;
    }
  }
  return fail(h, -5, "unknown op kind %d", op.d.kind);
}

int run_segment(ian_handle* h, int seg, int n, hipStream_t st) {
  ++h->run_serial;
  const bool fused = seg == IAN_SEG_DEC && head_fused_active(h, n);
  const HeadPlan& P = h->head;
  for (int i = 0; i < (int)h->ops.size(); ++i) {
    OpPlan& op = h->ops[i];
    if (op.d.segment != seg) continue;
    if (fused && (i == P.opR || i == P.opGa || i == P.opGb || i == P.opBa || i == P.opBb || i == P.opCat || i == P.opBeta)) {
      if (i == P.first) {   // the 128-channel map all of them hang off is ready: the whole head in two launches
        int rc = run_head_fused(h, n, st);
        if (rc) return rc;
      }
      continue;
    }
    int rc = run_op_fwd(h, op, n, st);
    if (rc) return rc;
  }
  return 0;
}

// ----- autotuner: per (layer, batch) pick tile shape x split-K policy by timing the real launch ------------
// The 9/6/6/4-tap parity classes, the 4^2..64^2 spatial extents and batch sizes from 1 to 1024 make the best
// (tile, K-range) decomposition layer specific; candidates are timed with HIP events on the caller's stream.
struct TuneCand {
  TgChoice ch;
  float ms;
};

std::vector<TgChoice> tune_candidates(const TgLayer& L, int nimg) {
  const int M = nimg * L.QH * L.QW;
  std::vector<int> cfgs;
  if (L.Cout <= 32) cfgs = {TG_128x32};
  else if (M <= 32) cfgs = {TG_32x128, TG_64x64};
  else if (M <= 64) cfgs = {TG_64x64, TG_32x128, TG_128x64};
  else {
    cfgs = {TG_128x128, TG_128x64, TG_64x64};
    if (M >= 512 && L.Cout >= 128) cfgs.push_back(TG_256x128);
    if (M >= 512 && L.Cout >= 128) cfgs.push_back(TG_128x128W8);
    if (M >= 512 && L.Cout >= 64) cfgs.push_back(TG_128x64W8);
  }
  int max_ksteps = 0;
  for (auto& c : L.classes) max_ksteps = std::max(max_ksteps, c.ntaps * (L.Cin / 32));
  std::vector<TgChoice> out;
  for (int cfg : cfgs) {
    const TgShape sh = tg_shape(cfg);
    const long long tiles = (long long)((M + sh.bm - 1) / sh.bm) * ((L.Cout + sh.bn - 1) / sh.bn);
    for (int ms : {0, 4, 8, 16, 32, 64, 128, 256}) {
      if (ms > 0 && ms >= max_ksteps) continue;  // would not split anything
      if (ms > 0) {
        long long slabs = 0;
        for (auto& c : L.classes) slabs += tiles * ((c.ntaps * (L.Cin / 32) + ms - 1) / ms);
        if (slabs * sh.bm * sh.bn * 4 > (512ll << 20)) continue;  // slab workspace cap
        if (slabs > 16384) continue;
      }
      for (int var : {1, 2, 4}) {   // 3 (LDS-DMA staging) measured 5 % slower on every 5x5 layer: selectable, not a candidate
        if (var == 4 && (M > 1024 || cfg == TG_256x128)) continue;   // the three-deep load queue is for latency-bound items (few images)
        TgChoice c;
        c.cfg = cfg;
        c.max_steps = ms;
        c.variant = var;
        out.push_back(c);
      }
    }
  }
  return out;
}

template <typename F>
int time_launches(ian_handle* h, hipStream_t st, int reps, F&& fn, float* ms) {
  hipEvent_t a, b;
  HIPCHK(h, hipEventCreate(&a));
  HIPCHK(h, hipEventCreate(&b));
  int rc = fn();  // warm-up (also builds + uploads the schedule)
  if (!rc) {
    (void)hipEventRecord(a, st);
    for (int r = 0; r < reps && !rc; ++r) rc = fn();
    (void)hipEventRecord(b, st);
    (void)hipEventSynchronize(b);
    float t = 0;
    (void)hipEventElapsedTime(&t, a, b);
    *ms = t / reps;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return rc;
}

template <typename F>
int tune_layer(ian_handle* h, TgLayer& L, int nimg, hipStream_t st, F&& run, TgChoice* best_out, float* best_ms) {
  TgChoice best;
  float bms = 1e30f;
  for (const TgChoice& c : tune_candidates(L, nimg)) {
    L.choice[nimg] = c;
    free_schedule_for(L, nimg);
    float ms = 0;
    int rc = time_launches(h, st, 5, run, &ms);
    if (rc) return rc;
    if (ms < bms) {
      bms = ms;
      best = c;
    }
  }
  L.choice[nimg] = best;
  free_schedule_for(L, nimg);
  if (best_out) *best_out = best;
  if (best_ms) *best_ms = bms;
  return 0;
}

// ----- autotune cache: IAN_TUNE_CACHE=<file> makes ian_autotune reuse / record its per-(batch, direction, layer)
// choices, so that repeated processes (e.g. the rocprofv3 passes of one profile) run the very same kernels.
typedef std::map<std::string, TgChoice> TuneCache;
std::string tune_key(int n, const char* dir, const std::string& name) { return std::to_string(n) + " " + dir + " " + name; }
void tune_cache_load(TuneCache& c) {
  const char* path = getenv("IAN_TUNE_CACHE");
  if (!path) return;
  FILE* f = fopen(path, "r");
  if (!f) return;
  char dir[16], name[256];
  int n, cfg, ms, var;
  while (fscanf(f, "%d %15s %255s %d %d %d", &n, dir, name, &cfg, &ms, &var) == 6) {
    TgChoice ch;
    ch.cfg = cfg;
    ch.max_steps = ms;
    ch.variant = var;
    c[tune_key(n, dir, name)] = ch;
  }
  fclose(f);
}
void tune_cache_store(const TuneCache& c) {
  const char* path = getenv("IAN_TUNE_CACHE");
  if (!path) return;
  FILE* f = fopen(path, "w");
  if (!f) return;
  for (auto& kv : c) fprintf(f, "%s %d %d %d\n", kv.first.c_str(), kv.second.cfg, kv.second.max_steps, kv.second.variant);
  fclose(f);
}

// stage a caller buffer (host or device) into a device pointer; returns pointer to use
int stage_in(ian_handle* h, const float* src, size_t count, float** buf, size_t* cap, const float** out,
             hipStream_t st) {
  if (is_device_ptr(src)) {
    *out = src;
    return 0;
  }
  if (count > *cap) {
    if (*buf) HIPCHK(h, hipFree(*buf));
    HIPCHK(h, hipMalloc((void**)buf, count * sizeof(float)));
    *cap = count;
    ++h->alloc_epoch;
  }
  HIPCHK(h, hipMemcpyAsync(*buf, src, count * sizeof(float), hipMemcpyHostToDevice, st));
  *out = *buf;
  return 0;
}

// A caller's DEVICE buffer stands in for an external-layout slot (l_in / l_out, NCHW) for the duration of one call: the first
// conv reads the images where they lie and the last op writes the reconstruction where it is wanted -- no boundary copy
// (2 x 3.1 MB D2D copies = 9 us of the 1.44 ms batch-64 step).  The slot's own buffer is left stale and marked so:
// ian_read_slot refuses it until a host-pointer call refills it.
struct SlotAlias {
  Slot* s = nullptr;
  float* saved_d = nullptr;
  size_t saved_cap = 0;
  void bind(ian_handle* h, int slot, const float* p, int n) {
    s = &h->slots[slot];
    saved_d = s->d;
    saved_cap = s->cap;
    s->d = const_cast<float*>(p);
    s->cap = s->per_image() * (size_t)n;
    h->slot_stale[slot] = 1;
  }
  ~SlotAlias() {
    if (s) {
      s->d = saved_d;
      s->cap = saved_cap;
    }
  }
};

int set_image_input(ian_handle* h, const float* x, int n, hipStream_t st) {
  Slot& xs = h->slots[h->desc.x_slot];
  const size_t count = xs.per_image() * (size_t)n;
  int rc;
  if ((rc = ensure_slot(h, h->desc.x_slot, n))) return rc;
  HIPCHK(h, hipMemcpyAsync(xs.d, x, count * sizeof(float), is_device_ptr(x) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  h->slot_stale[h->desc.x_slot] = 0;
  return 0;
}

int set_latent_input(ian_handle* h, int slot, const float* z, int n, hipStream_t st) {
  Slot& zs = h->slots[slot];
  int rc;
  if ((rc = ensure_slot(h, slot, n))) return rc;
  const float* dz;
  if ((rc = stage_in(h, z, (size_t)n * zs.c, &h->d_stage_in, &h->stage_in_cap, &dz, st))) return rc;
  HIPCHK(h, launch_rows_copy(dz, zs.c, zs.d, zs.cs, n, zs.c, st));
  return 0;
}

int get_latent_output(ian_handle* h, int slot, float* z, int n, hipStream_t st) {
  Slot& zs = h->slots[slot];
  if (is_device_ptr(z)) {
    HIPCHK(h, launch_rows_copy(zs.d, zs.cs, z, zs.c, n, zs.c, st));
    return 0;
  }
  const size_t count = (size_t)n * zs.c;
  if (count > h->stage_out_cap) {
    if (h->d_stage_out) HIPCHK(h, hipFree(h->d_stage_out));
    HIPCHK(h, hipMalloc((void**)&h->d_stage_out, count * sizeof(float)));
    h->stage_out_cap = count;
    ++h->alloc_epoch;
  }
  HIPCHK(h, launch_rows_copy(zs.d, zs.cs, h->d_stage_out, zs.c, n, zs.c, st));
  HIPCHK(h, hipMemcpyAsync(z, h->d_stage_out, count * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  h->last_pending = false;
  return 0;
}

int get_image_output(ian_handle* h, float* x, int n, hipStream_t st) {
  Slot& os = h->slots[h->desc.out_slot];
  h->slot_stale[h->desc.out_slot] = 0;
  const size_t count = os.per_image() * (size_t)n;
  const bool dev = is_device_ptr(x);
  HIPCHK(h, hipMemcpyAsync(x, os.d, count * sizeof(float), dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
  if (!dev) {
    HIPCHK(h, hipStreamSynchronize(st));
    h->last_pending = false;
  }
  return 0;
}

int check_ready(ian_handle* h, int n) {
  if (!h) return -1;
  if (!h->finalized) return fail(h, -6, "ian_finalize has not been called");
  if (n <= 0) return fail(h, -7, "batch size must be positive (got %d)", n);
  return 0;
}

struct TotalTimer {  // whole-call device time when profiling
  ian_handle* h;
  hipStream_t st;
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  TotalTimer(ian_handle* h_, hipStream_t s) : h(h_), st(s) {
    if (!h->prof) return;
    if (h->ev_total_used == h->ev_total.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      h->ev_total.push_back({a, b});
    }
    ev = &h->ev_total[h->ev_total_used++];
    (void)hipEventRecord(ev->first, st);
  }
  ~TotalTimer() {
    if (ev) (void)hipEventRecord(ev->second, st);
  }
};

// ----- latent-brush backward (API.py:59,64): reverse sweep over the decoder ops --------------------------
// Gradient buffers (Slot::g) hold dL/d(pre-epilogue value) of the slot's producer, so each backward tapgemm's
// epilogue (TG_EPI_BWD) multiplies by act'(y)*scale of the producer of ITS output slot and no separate
// elementwise pass runs on the GEMM edges.  The decoder graph of the full IAN is a DAG (residual adds, the
// shared feature map of the RGB-Beta head, concat): the ops are visited in reverse topological order and a slot
// that has more than one consumer accumulates (the epilogue's `res` input / the accumulate flag of the
// identity-edge kernels), "touched" recording whether a contribution already arrived in this sweep.
struct ProducerEpi {
  const float* scale = nullptr;
  const float* yfwd = nullptr;
  int act = IAN_ACT_NONE;
  int scale_period = 0;
};

int run_decoder_backward(ian_handle* h, int mode, int c1, int r1, int c2, int r2, const float* d_rgb, hipStream_t st,
                         const int* d_patch = nullptr) {
  std::vector<OpPlan*> dec;
  for (auto& op : h->ops)
    if (op.d.segment == IAN_SEG_DEC) dec.push_back(&op);
  if (dec.empty()) return fail(h, -8, "no decoder ops");
  const int nslots = (int)h->slots.size();
  std::vector<OpPlan*> prod(nslots, nullptr);
  for (OpPlan* op : dec) prod[op->d.dst] = op;
  OpPlan& last = *dec.back();
  if (last.d.dst != h->desc.out_slot) return fail(h, -9, "imgrad: the last decoder op does not produce l_out");
  Slot& out = h->slots[last.d.dst];
  const int H = out.h, W = out.w;
  if (c1 < 0 || r1 < 0 || c2 > W || r2 > H) return fail(h, -7, "patch (%d,%d,%d,%d) outside the %dx%d image", c1, r1, c2, r2, W, H);
  if (!h->d_gseed) {
    HIPCHK(h, hipMalloc((void**)&h->d_gseed, (size_t)3 * H * W * sizeof(float)));
    ++h->alloc_epoch;
  }
  // interactive loop + image-producing deconv (IAN_simple): seed, tanh' and the first backward-data in one launch
  const bool fused_seed = d_patch && last.d.kind == IAN_OP_DECONV5S2 && last.edge;
  if (fused_seed) {
  } else if (d_patch) HIPCHK(h, launch_patch_seed_dev(out.d, d_rgb, h->d_gseed, H, W, d_patch, mode, st));
  else HIPCHK(h, launch_patch_seed(out.d, d_rgb, h->d_gseed, H, W, c1, r1, c2, r2, mode, st));  // dL/dX_hat, NCHW

  std::vector<char> touched(nslots, 0);
  auto epi_of = [&](int slot) {  // what turns a value-gradient of `slot` into its producer's pre-epilogue gradient
    ProducerEpi e;
    OpPlan* p = prod[slot];
    if (!p) return e;
    Slot& s = h->slots[slot];
    e.scale = p->d_scale;
    e.act = p->d.act;
    e.yfwd = (e.act != IAN_ACT_NONE) ? s.d : nullptr;
    // a dense producer's batch-norm is per feature = per (pixel, channel) of the map it is reshaped to
    e.scale_period = (p->d.kind == IAN_OP_DENSE && s.h * s.w > 1) ? (int)s.per_image() : 0;
    return e;
  };
  auto pass_to = [&](const float* gs, int ss, int coff, int slot, int C) -> int {
    Slot& t = h->slots[slot];
    int rc = ensure_slot(h, slot, 1, true);
    if (rc) return rc;
    ProducerEpi e = epi_of(slot);
    if (e.scale_period) return fail(h, -9, "imgrad: identity edge into a per-feature batch-norm is not supported");
    HIPCHK(h, launch_grad_pass(gs, ss, coff, t.g, e.yfwd, t.cs, e.scale, (long long)t.h * t.w, C, e.act, touched[slot], st));
    touched[slot] = 1;
    return 0;
  };

  for (int i = (int)dec.size() - 1; i >= 0; --i) {
    OpPlan& op = *dec[i];
    const int kind = op.d.kind;
    int rc;
    if (kind == IAN_OP_BETA) {  // IAN.py:207
      const int srcs[3] = {op.d.src, op.d.src2, op.d.src3};
      BetaBwdArgs a;
      for (int c = 0; c < 3; ++c) {
        Slot& m = h->slots[srcs[c]];
        if ((rc = ensure_slot(h, srcs[c], 1, true))) return rc;
        ProducerEpi e = epi_of(srcs[c]);
        a.v[c] = m.d; a.g[c] = m.g; a.scale[c] = e.scale; a.act[c] = e.act; a.accumulate[c] = touched[srcs[c]];
        touched[srcs[c]] = 1;
      }
      Slot& m0 = h->slots[srcs[0]];
      HIPCHK(h, launch_beta_bwd(h->d_gseed, a, 1, m0.h * m0.w, m0.cs, st));
      continue;
    }
    if (kind == IAN_OP_DECONV5S2 && op.edge) {  // image-producing deconv (IAN_simple dec_out)
      if (op.d.dst != h->desc.out_slot) return fail(h, -9, "imgrad: edge deconv '%s' must produce l_out", op.name.c_str());
      Slot& in = h->slots[op.d.src];
      if (touched[op.d.src]) return fail(h, -9, "imgrad: '%s' input has several consumers", op.name.c_str());
      if ((rc = ensure_slot(h, op.d.src, 1, true))) return rc;
      ProducerEpi e = epi_of(op.d.src);
      if (e.scale_period) return fail(h, -9, "imgrad: per-feature batch-norm directly under the image deconv");
      if (fused_seed) {
        HIPCHK(h, launch_deconv_out_bwd_seed(out.d, d_rgb, d_patch, mode, op.d.act, op.d_scale, op.d_edge_w, in.g, e.yfwd, e.scale,
                                             op.d.in_h, op.d.in_w, in.cs, op.d.cout, e.act, st));
      } else {
        HIPCHK(h, launch_dact_nchw(h->d_gseed, out.d, op.d_scale, 1, out.c, H * W, op.d.act, st));
        HIPCHK(h, launch_deconv_out_bwd(h->d_gseed, op.d_edge_w, in.g, e.yfwd, e.scale, 1, op.d.in_h, op.d.in_w, in.cs,
                                        op.d.cout, e.act, st));
      }
      touched[op.d.src] = 1;
      continue;
    }
    if (!touched[op.d.dst]) return fail(h, -9, "imgrad: no gradient reached the output of '%s'", op.name.c_str());
    Slot& o = h->slots[op.d.dst];
    switch (kind) {
      case IAN_OP_DENSE:
      case IAN_OP_DECONV5S2:
      case IAN_OP_MDC3: {
        if (!op.bwd.valid) return fail(h, -9, "imgrad: op '%s' has no backward-data form", op.name.c_str());
        Slot& in = h->slots[op.d.src];
        if ((rc = ensure_slot(h, op.d.src, 1, true))) return rc;
        ProducerEpi pe = epi_of(op.d.src);
        TgEpilogue e;
        e.scale = pe.scale; e.shift = nullptr; e.yfwd = pe.yfwd; e.act = pe.act; e.scale_period = pe.scale_period;
        e.res = touched[op.d.src] ? in.g : nullptr;
        e.mode = TG_EPI_BWD;
        const int ystride = (kind == IAN_OP_DENSE) ? (int)in.per_image() : in.cs;
        if (kind == IAN_OP_DENSE && h->opt.dense_gemv && !pe.scale && !pe.yfwd && pe.act == IAN_ACT_NONE && op.d.flat_c <= 0 &&
            (op.bwd.Cin & 3) == 0 && op.bwd.Cout <= ystride) {
          // the latent's own layer: slab [in][out], row j contiguous in the (permuted) output index = o.g's order
          HIPCHK(h, launch_dense_bwd_gemv(o.g, op.bwd.d_w, op.bwd.Cout, op.bwd.Cin, e.res, in.g, st));
        } else {
          B1Params bp;
          if (kind == IAN_OP_DECONV5S2 && b1_fill(h, op, 1, o, in, bp)) {
            bp.x = o.g; bp.y = in.g; bp.scale = e.scale; bp.yfwd = e.yfwd; bp.res = e.res; bp.act = e.act; bp.bwd = 1;
            bp.scale_period = e.scale_period;
            HIPCHK(h, launch_b1conv(bp, 1, st));
          } else if ((rc = run_tapgemm(h, op.bwd, 1, o.g, in.g, ystride, e, st))) return rc;
        }
        touched[op.d.src] = 1;
        if (kind == IAN_OP_MDC3 && op.d.src2 >= 0)  // residual operand of the fused ElemwiseSum: identity edge
          if ((rc = pass_to(o.g, o.cs, 0, op.d.src2, o.c))) return rc;
        break;
      }
      case IAN_OP_AFFINE:  // stand-alone BatchNorm(+nonlinearity): its g already is dL/d(input value)
        if ((rc = pass_to(o.g, o.cs, 0, op.d.src, o.c))) return rc;
        break;
      case IAN_OP_CONCAT: {
        const int ca = h->slots[op.d.src].c, cb = h->slots[op.d.src2].c;
        if ((rc = pass_to(o.g, o.cs, 0, op.d.src, ca))) return rc;
        if ((rc = pass_to(o.g, o.cs, ca, op.d.src2, cb))) return rc;
        break;
      }
      default:
        return fail(h, -9, "imgrad: backward of op kind %d ('%s') is not implemented", kind, op.name.c_str());
    }
  }
  if (!touched[h->desc.z_slot]) return fail(h, -9, "imgrad: no gradient reached the latent");
  return 0;
}

// ----- stream hand-over between calls ------------------------------------------------------------------------
// The handle's buffers are shared by every call; a call that returns without synchronising (device output pointers)
// leaves work pending on its stream.  When the next call runs on ANOTHER stream, wait for that work first.
void enter_stream(ian_handle* h, hipStream_t st) {
  if (h->last_pending && h->last_stream != st) (void)hipStreamSynchronize(h->last_stream);
  h->last_stream = st;
  h->last_pending = true;   // cleared by callers that synchronise before returning
}

// ----- captured graphs for the interactive loop ----------------------------------------------------------------
constexpr int PIN_Z = 0, PIN_DZ = 128, PIN_IMG = 256, PIN_FLOATS = 256 + 3 * 64 * 64 + 8;

bool edit_graph_eligible(const ian_handle* h, void* stream, std::initializer_list<const void*> host_ptrs) {
  if (!h->opt.edit_graph || h->prof || stream != nullptr || h->graph_failed || h->desc.num_latents > 128) return false;
  const Slot& os = h->slots[h->desc.out_slot];
  if (os.per_image() > 3 * 64 * 64) return false;
  for (const void* p : host_ptrs)
    if (p && is_device_ptr(p)) return false;
  return true;
}

int edit_ctx(ian_handle* h) {
  if (h->edit_stream) return 0;
  HIPCHK(h, hipStreamCreateWithFlags(&h->edit_stream, hipStreamNonBlocking));
  HIPCHK(h, hipHostMalloc((void**)&h->pin, PIN_FLOATS * sizeof(float), hipHostMallocDefault));
  HIPCHK(h, hipMalloc((void**)&h->d_patch, 8 * sizeof(int)));   // brush rectangle + (coef, gscale) of ian_brush_step
  if (128 > h->stage_in_cap) {
    if (h->d_stage_in) HIPCHK(h, hipFree(h->d_stage_in));
    HIPCHK(h, hipMalloc((void**)&h->d_stage_in, 128 * sizeof(float)));
    h->stage_in_cap = 128;
  }
  if (128 > h->stage_out_cap) {
    if (h->d_stage_out) HIPCHK(h, hipFree(h->d_stage_out));
    HIPCHK(h, hipMalloc((void**)&h->d_stage_out, 128 * sizeof(float)));
    h->stage_out_cap = 128;
  }
  ++h->alloc_epoch;
  return 0;
}

// Run `body` (a fixed sequence of launches / async copies on h->edit_stream): eagerly the first times (allocations,
// schedule uploads, function attributes happen there), then captured once, then replayed.
template <typename F>
int run_or_replay(ian_handle* h, ian_handle::EditGraph& G, F&& body) {
  hipStream_t st = h->edit_stream;
  if (G.exec && G.epoch == h->alloc_epoch) {
    HIPCHK(h, hipGraphLaunch(G.exec, st));
    return 0;
  }
  if (G.exec) {
    (void)hipGraphExecDestroy(G.exec);
    G.exec = nullptr;
    G.warm = 0;
  }
  if (G.warm < 1 || G.warm_epoch != h->alloc_epoch || h->graph_failed) {
    // a capture must replay exactly what an eager pass has already done once: first launches set function attributes,
    // build and upload schedules, allocate -- none of which may happen inside a capture.  An option change or a
    // (re)allocation between the eager pass and the capture (alloc_epoch moved) therefore asks for another eager pass.
    const long long e0 = h->alloc_epoch;
    const int rc = body();
    G.warm = (h->alloc_epoch == e0) ? 1 : 0;
    G.warm_epoch = h->alloc_epoch;
    return rc;
  }
  const long long e0 = h->alloc_epoch;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    h->graph_failed = true;
    return body();
  }
  const int rc = body();
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(st, &g);
  if (rc || e != hipSuccess || !g || h->alloc_epoch != e0) {   // never replay a graph whose capture hit an error or an allocation
    (void)hipGetLastError();
    if (g) (void)hipGraphDestroy(g);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {   // an invalidated capture may still hold the stream
      hipGraph_t g2 = nullptr;
      (void)hipStreamEndCapture(st, &g2);
      if (g2) (void)hipGraphDestroy(g2);
    }
    (void)hipGetLastError();
    h->graph_failed = true;
    h->err.clear();
    return body();
  }
  const hipError_t ei = hipGraphInstantiate(&G.exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ei != hipSuccess) {
    (void)hipGetLastError();
    G.exec = nullptr;
    h->graph_failed = true;
    return body();
  }
  G.epoch = h->alloc_epoch;
  HIPCHK(h, hipGraphLaunch(G.exec, st));
  return 0;
}

bool dec_cache_hit(const ian_handle* h, const float* z) {
  return h->dec_cache_valid && !is_device_ptr(z) && getenv("IAN_NO_DEC_CACHE") == nullptr &&
         memcmp(h->dec_cache_z.data(), z, sizeof(float) * h->desc.num_latents) == 0;
}

// batch-1 decoder forward for the HOST latent z on the internal stream (graph replay), unless the resident activations
// already belong to it
int decode_one_graph(ian_handle* h, const float* z) {
  if (dec_cache_hit(h, z)) return 0;
  h->dec_cache_valid = false;
  int rc;
  if ((rc = ensure_slot(h, h->desc.z_slot, 1))) return rc;
  const int zl = h->desc.num_latents;
  memcpy(h->pin + PIN_Z, z, zl * sizeof(float));
  hipStream_t st = h->edit_stream;
  rc = run_or_replay(h, h->g_fwd, [&]() -> int {
    Slot& zs = h->slots[h->desc.z_slot];
    // one row: straight into the slot (its channel padding beyond num_latents stays zero), no staging kernel
    HIPCHK(h, hipMemcpyAsync(zs.d, h->pin + PIN_Z, zl * sizeof(float), hipMemcpyHostToDevice, st));
    return run_segment(h, IAN_SEG_DEC, 1, st);
  });
  if (rc) return rc;
  h->dec_cache_z.assign(z, z + zl);
  h->dec_cache_valid = true;
  return 0;
}

// batch-1 decoder forward for latent z unless the resident activations already belong to it (NPE.py:205,218:
// imgradRGB(z) right after sample_at(z), the blend right after the brush step)
int decode_one_cached(ian_handle* h, const float* z, hipStream_t st) {
  int rc;
  const bool host_z = !is_device_ptr(z);
  if (!dec_cache_hit(h, z)) {
    h->dec_cache_valid = false;
    if ((rc = set_latent_input(h, h->desc.z_slot, z, 1, st))) return rc;
    if ((rc = run_segment(h, IAN_SEG_DEC, 1, st))) return rc;
    if (host_z) {
      h->dec_cache_z.assign(z, z + h->desc.num_latents);
      h->dec_cache_valid = true;
    }
  }
  return 0;
}

int upload_rgb_if_changed(ian_handle* h, const float* rgb, hipStream_t st, const float** d_rgb) {
  Slot& out = h->slots[h->desc.out_slot];
  const size_t cnt = out.per_image();
  if (is_device_ptr(rgb)) {
    *d_rgb = rgb;
    return 0;
  }
  if (!h->d_rgb) {
    HIPCHK(h, hipMalloc((void**)&h->d_rgb, cnt * sizeof(float)));
    ++h->alloc_epoch;
  }
  // the brush colour image rarely changes between motion events (NPE.py:205 passes the same myRGB): re-upload only when
  // the host bytes differ from the last upload
  if (h->rgb_cache.size() != cnt || memcmp(h->rgb_cache.data(), rgb, cnt * sizeof(float)) != 0) {
    h->rgb_cache.assign(rgb, rgb + cnt);   // the shadow copy is the upload source: it outlives the caller's buffer
    HIPCHK(h, hipMemcpyAsync(h->d_rgb, h->rgb_cache.data(), cnt * sizeof(float), hipMemcpyHostToDevice, st));
  }
  *d_rgb = h->d_rgb;
  return 0;
}

int grad_common(ian_handle* h, int mode, int c1, int r1, int c2, int r2, const float* rgb, const float* z, float* dz,
                void* stream) {
  int rc = check_ready(h, 1);
  if (rc) return rc;
  if (!z || !dz) return fail(h, -1, "null pointer passed to ian_grad_*");
  Slot& zs = h->slots[h->desc.z_slot];
  if (edit_graph_eligible(h, stream, {z, dz, rgb})) {
    Slot& out = h->slots[h->desc.out_slot];
    if (c1 < 0 || r1 < 0 || c2 > out.w || r2 > out.h) return fail(h, -7, "patch (%d,%d,%d,%d) outside the %dx%d image", c1, r1, c2, r2, out.w, out.h);
    if ((rc = edit_ctx(h))) return rc;
    hipStream_t st = h->edit_stream;
    enter_stream(h, st);
    if ((rc = decode_one_graph(h, z))) return rc;
    const float* d_rgb = nullptr;
    if (mode == 1 && (rc = upload_rgb_if_changed(h, rgb, st, &d_rgb))) return rc;
    if ((rc = ensure_slot(h, h->desc.z_slot, 1, true))) return rc;
    int* patch = reinterpret_cast<int*>(h->pin + PIN_IMG + 3 * 64 * 64);
    patch[0] = c1; patch[1] = r1; patch[2] = c2; patch[3] = r2;
    rc = run_or_replay(h, h->g_bwd[mode], [&]() -> int {
      HIPCHK(h, hipMemcpyAsync(h->d_patch, patch, 4 * sizeof(int), hipMemcpyHostToDevice, st));
      int r = run_decoder_backward(h, mode, 0, 0, out.w, out.h, d_rgb, st, h->d_patch);
      if (r) return r;
      HIPCHK(h, hipMemcpyAsync(h->pin + PIN_DZ, zs.g, zs.c * sizeof(float), hipMemcpyDeviceToHost, st));
      return 0;
    });
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(st));
    h->last_pending = false;
    memcpy(dz, h->pin + PIN_DZ, zs.c * sizeof(float));
    return 0;
  }
  hipStream_t st = (hipStream_t)stream;
  enter_stream(h, st);
  TotalTimer tt(h, st);
  if ((rc = decode_one_cached(h, z, st))) return rc;
  const float* d_rgb = nullptr;
  if (mode == 1 && (rc = upload_rgb_if_changed(h, rgb, st, &d_rgb))) return rc;
  if ((rc = run_decoder_backward(h, mode, c1, r1, c2, r2, d_rgb, st))) return rc;
  // result sits in the gradient buffer of the z slot
  if (is_device_ptr(dz)) {
    HIPCHK(h, launch_rows_copy(zs.g, zs.cs, dz, zs.c, 1, zs.c, st));
  } else {
    if ((size_t)zs.c > h->stage_out_cap) {
      if (h->d_stage_out) HIPCHK(h, hipFree(h->d_stage_out));
      HIPCHK(h, hipMalloc((void**)&h->d_stage_out, zs.c * sizeof(float)));
      h->stage_out_cap = zs.c;
      ++h->alloc_epoch;
    }
    HIPCHK(h, launch_rows_copy(zs.g, zs.cs, h->d_stage_out, zs.c, 1, zs.c, st));
    HIPCHK(h, hipMemcpyAsync(dz, h->d_stage_out, zs.c * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    h->last_pending = false;
  }
  return 0;
}

// NPE.py:218-231 for the image resident in the output slot: (cached) uploads of RECON / ERROR, the blend kernel, copies
// back.  *pending = a host copy was enqueued (the caller synchronises).
int enqueue_photo_blend(ian_handle* h, const uint8_t* recon, const float* error, const double* gauss_half, int radius, uint8_t* im,
                        double* mask, hipStream_t st, bool* pending) {
  if (!recon || !error || !gauss_half || !im) return fail(h, -1, "null pointer passed to the photo blend");
  if (radius < 0 || radius > 7) return fail(h, -7, "photo blend: radius %d outside 0..7", radius);
  Slot& os = h->slots[h->desc.out_slot];
  if (os.h != 64 || os.w != 64 || os.c != 3) return fail(h, -7, "the photo blend needs a 3x64x64 image");
  const size_t cnt = 3 * 64 * 64;
  PhotoBlendArgs a;
  memset(&a, 0, sizeof a);
  a.xhat = os.d;
  if (is_device_ptr(recon)) a.recon = recon;
  else {
    if (!h->d_recon) HIPCHK(h, hipMalloc((void**)&h->d_recon, cnt));
    if (h->recon_cache.size() != cnt || memcmp(h->recon_cache.data(), recon, cnt) != 0) {   // changes on infer / Reset only
      h->recon_cache.assign(recon, recon + cnt);
      HIPCHK(h, hipMemcpyAsync(h->d_recon, h->recon_cache.data(), cnt, hipMemcpyHostToDevice, st));
    }
    a.recon = h->d_recon;
  }
  if (is_device_ptr(error)) a.error = error;
  else {
    if (!h->d_error) HIPCHK(h, hipMalloc((void**)&h->d_error, cnt * sizeof(float)));
    if (h->error_cache.size() != cnt || memcmp(h->error_cache.data(), error, cnt * sizeof(float)) != 0) {
      h->error_cache.assign(error, error + cnt);
      HIPCHK(h, hipMemcpyAsync(h->d_error, h->error_cache.data(), cnt * sizeof(float), hipMemcpyHostToDevice, st));
    }
    a.error = h->d_error;
  }
  const bool im_dev = is_device_ptr(im), mask_dev = mask && is_device_ptr(mask);
  if (!im_dev && !h->d_im) HIPCHK(h, hipMalloc((void**)&h->d_im, cnt));
  if (mask && !mask_dev && !h->d_mask) HIPCHK(h, hipMalloc((void**)&h->d_mask, 64 * 64 * sizeof(double)));
  a.im = im_dev ? im : h->d_im;
  a.mask = mask ? (mask_dev ? mask : h->d_mask) : nullptr;
  for (int i = 0; i <= radius; ++i) a.w[i] = gauss_half[i];
  a.radius = radius;
  HIPCHK(h, launch_photo_blend(a, st));
  if (!im_dev) HIPCHK(h, hipMemcpyAsync(im, h->d_im, cnt, hipMemcpyDeviceToHost, st));
  if (mask && !mask_dev) HIPCHK(h, hipMemcpyAsync(mask, h->d_mask, 64 * 64 * sizeof(double), hipMemcpyDeviceToHost, st));
  *pending = !im_dev || (mask && !mask_dev);
  return 0;
}


}  // namespace

// =========================================== C ABI ====================================================
extern "C" {

const char* ian_version(void) { return "libian 0.1 (gfx950)"; }

const char* ian_last_error(ian_handle* h) { return h ? h->err.c_str() : "null handle"; }

int ian_create(const ian_model_desc* desc, ian_handle** out) {
  if (!desc || !out) return -1;
  std::unique_ptr<ian_handle> h(new ian_handle());
  apply_env_options(h->opt);
  h->desc = *desc;
  h->strings.reserve((size_t)desc->n_ops * 2 + 4);
  h->ops.resize(desc->n_ops);
  for (int i = 0; i < desc->n_ops; ++i) {
    OpPlan& op = h->ops[i];
    op.d = desc->ops[i];
    op.name = op.d.name ? op.d.name : "";
    op.bn_name = op.d.bn_name ? op.d.bn_name : "";
    op.d.name = nullptr;
    op.d.bn_name = nullptr;
    if (op.d.n_scales < 0 || op.d.n_scales > IAN_MAX_SCALES) return -1;
    auto bad = [&](int s) { return s < -1 || s >= desc->n_slots; };
    if (bad(op.d.src) || bad(op.d.src2) || bad(op.d.src3) || bad(op.d.dst) || op.d.src < 0 || op.d.dst < 0) return -1;
  }
  h->slots.resize(desc->n_slots);
  for (int i = 0; i < desc->n_slots; ++i) {
    Slot& s = h->slots[i];
    s.h = desc->slots[i].h; s.w = desc->slots[i].w; s.c = desc->slots[i].c;
    s.cs = round_up(s.c, 32);
    s.nchw = (i == desc->x_slot || i == desc->out_slot);
  }
  h->desc.ops = nullptr;
  h->desc.slots = nullptr;
  *out = h.release();
  return 0;
}

int ian_load_param(ian_handle* h, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
  if (!h || !name || !data || ndim < 0 || ndim > 8) return fail(h, -1, "bad argument to ian_load_param");
  if (h->finalized) return fail(h, -6, "model already finalized");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  const int64_t n = t.numel();
  t.data.assign(data, data + n);
  h->params[name] = std::move(t);
  return 0;
}

int ian_set_made_masks(ian_handle* h, const float* m0, const float* m1, const float* md, int32_t n) {
  if (!h || !m0 || !m1 || !md || n <= 0) return fail(h, -1, "bad argument to ian_set_made_masks");
  const float* src[3] = {m0, m1, md};
  for (int k = 0; k < 3; ++k) {
    h->made_masks[k].assign(src[k], src[k] + (size_t)n * n);
    for (float v : h->made_masks[k])
      if (v != 0.f && v != 1.f) return fail(h, -3, "MADE mask %d is not 0/1 valued", k);
  }
  h->made_n = n;
  return 0;
}

int ian_finalize(ian_handle* h) {
  if (!h) return -1;
  if (h->finalized) return fail(h, -6, "model already finalized");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return fail(h, -10, "no HIP device available: libian has no CPU fallback");
  }
  for (auto& op : h->ops) {
    int rc = 0;
    std::vector<int> out_perm;
    bool has_out_perm = false;
    switch (op.d.kind) {
      case IAN_OP_CONV5S2:
        if ((rc = pack_conv_fwd(h, op))) return rc;
        if ((rc = build_affine(h, op, op.d.cout, nullptr))) return rc;
        break;
      case IAN_OP_DECONV5S2:
        if ((rc = pack_deconv(h, op))) return rc;
        if ((rc = build_affine(h, op, op.d.cout, nullptr))) return rc;
        break;
      case IAN_OP_MDC3:
        if ((rc = pack_mdc(h, op))) return rc;
        if ((rc = build_affine(h, op, op.d.cout, nullptr))) return rc;
        break;
      case IAN_OP_DENSE:
        if ((rc = pack_dense(h, op, out_perm, has_out_perm))) return rc;
        if ((rc = build_affine(h, op, op.d.cout, has_out_perm ? &out_perm : nullptr))) return rc;
        break;
      case IAN_OP_AFFINE:
        if ((rc = build_affine(h, op, op.d.cout, nullptr))) return rc;
        break;
      case IAN_OP_MADE_IAF:
        if ((rc = pack_made(h, op))) return rc;
        break;
      case IAN_OP_BETA:
      case IAN_OP_CONCAT:
        break;
      default:
        return fail(h, -5, "unknown op kind %d", op.d.kind);
    }
    if ((rc = upload_layer(h, op.fwd))) return rc;
    if ((rc = upload_layer(h, op.bwd))) return rc;
    if ((rc = upload(h, op.h_edge_w, &op.d_edge_w))) return rc;
    if ((rc = upload(h, op.h_b1_fwd, &op.d_b1_fwd))) return rc;
    if ((rc = upload(h, op.h_b1_bwd, &op.d_b1_bwd))) return rc;
    std::vector<float>().swap(op.h_b1_fwd);
    std::vector<float>().swap(op.h_b1_bwd);
    if ((rc = upload(h, op.h_scale, &op.d_scale))) return rc;
    if ((rc = upload(h, op.h_shift, &op.d_shift))) return rc;
    std::vector<float>().swap(op.h_edge_w);
  }
  h->params.clear();
  h->finalized = true;
  return find_head_plan(h);
}

int ian_encode_pre_iaf(ian_handle* h, const float* x, int32_t n, float* z, void* stream) {
  int rc = check_ready(h, n);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  enter_stream(h, st);
  TotalTimer tt(h, st);
  if ((rc = set_image_input(h, x, n, st))) return rc;
  if ((rc = run_segment(h, IAN_SEG_ENC, n, st))) return rc;
  return get_latent_output(h, h->desc.zpre_slot, z, n, st);
}

int ian_encode(ian_handle* h, const float* x, int32_t n, float* z, void* stream) {
  int rc = check_ready(h, n);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  enter_stream(h, st);
  TotalTimer tt(h, st);
  h->dec_cache_valid = false;
  if ((rc = set_image_input(h, x, n, st))) return rc;
  if ((rc = run_segment(h, IAN_SEG_ENC, n, st))) return rc;
  if ((rc = run_segment(h, IAN_SEG_IAF, n, st))) return rc;
  return get_latent_output(h, h->desc.z_slot, z, n, st);
}

int ian_iaf(ian_handle* h, const float* zpre, int32_t n, float* z, void* stream) {
  int rc = check_ready(h, n);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  enter_stream(h, st);
  h->dec_cache_valid = false;
  if ((rc = set_latent_input(h, h->desc.zpre_slot, zpre, n, st))) return rc;
  if ((rc = run_segment(h, IAN_SEG_IAF, n, st))) return rc;
  return get_latent_output(h, h->desc.z_slot, z, n, st);
}

int ian_decode(ian_handle* h, const float* z, int32_t n, float* x, void* stream) {
  int rc = check_ready(h, n);
  if (rc) return rc;
  if (!z || !x) return fail(h, -1, "null pointer passed to ian_decode");
  if (n == 1 && edit_graph_eligible(h, stream, {z, x})) {   // NPE.py:110,218,261: one latent, host buffers
    if ((rc = edit_ctx(h))) return rc;
    hipStream_t st = h->edit_stream;
    enter_stream(h, st);
    if ((rc = decode_one_graph(h, z))) return rc;
    Slot& os = h->slots[h->desc.out_slot];
    HIPCHK(h, hipMemcpyAsync(h->pin + PIN_IMG, os.d, os.per_image() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    h->last_pending = false;
    memcpy(x, h->pin + PIN_IMG, os.per_image() * sizeof(float));
    return 0;
  }
  hipStream_t st = (hipStream_t)stream;
  enter_stream(h, st);
  TotalTimer tt(h, st);
  h->dec_cache_valid = false;
  if ((rc = set_latent_input(h, h->desc.z_slot, z, n, st))) return rc;
  if ((rc = run_segment(h, IAN_SEG_DEC, n, st))) return rc;
  if ((rc = get_image_output(h, x, n, st))) return rc;
  if (n == 1 && !is_device_ptr(z)) {  // remember which latent the resident batch-1 activations belong to
    h->dec_cache_z.assign(z, z + h->desc.num_latents);
    h->dec_cache_valid = true;
  }
  return 0;
}

int ian_reconstruct(ian_handle* h, const float* x, int32_t n, float* xhat, void* stream) {
  int rc = check_ready(h, n);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  enter_stream(h, st);
  TotalTimer tt(h, st);
  h->dec_cache_valid = false;
  if (!x || !xhat) return fail(h, -1, "null pointer passed to ian_reconstruct");
  SlotAlias in_alias, out_alias;
  if (h->opt.alias_io && is_device_ptr(x) && h->slots[h->desc.x_slot].nchw) in_alias.bind(h, h->desc.x_slot, x, n);
  else if ((rc = set_image_input(h, x, n, st))) return rc;
  if ((rc = run_segment(h, IAN_SEG_ENC, n, st))) return rc;
  if ((rc = run_segment(h, IAN_SEG_IAF, n, st))) return rc;
  if (h->opt.alias_io && is_device_ptr(xhat) && h->slots[h->desc.out_slot].nchw && xhat != x) {
    out_alias.bind(h, h->desc.out_slot, xhat, n);
    return run_segment(h, IAN_SEG_DEC, n, st);
  }
  if ((rc = run_segment(h, IAN_SEG_DEC, n, st))) return rc;
  return get_image_output(h, xhat, n, st);
}

int ian_decode_u8(ian_handle* h, const float* z, int32_t n, uint8_t* out, void* stream) {
  int rc = check_ready(h, n);
  if (rc) return rc;
  if (!z || !out) return fail(h, -1, "null pointer passed to ian_decode_u8");
  hipStream_t st = (hipStream_t)stream;
  const bool graph = n == 1 && edit_graph_eligible(h, stream, {z, out});
  if (graph) {
    if ((rc = edit_ctx(h))) return rc;
    st = h->edit_stream;
  }
  enter_stream(h, st);
  TotalTimer tt(h, st);
  if (graph) {
    if ((rc = decode_one_graph(h, z))) return rc;
  } else if (n == 1) {
    if ((rc = decode_one_cached(h, z, st))) return rc;
  } else {
    h->dec_cache_valid = false;
    if ((rc = set_latent_input(h, h->desc.z_slot, z, n, st))) return rc;
    if ((rc = run_segment(h, IAN_SEG_DEC, n, st))) return rc;
  }
  Slot& os = h->slots[h->desc.out_slot];
  const size_t count = os.per_image() * (size_t)n;
  if (is_device_ptr(out)) {
    HIPCHK(h, launch_to_uint8(os.d, out, (long long)count, st));
    return 0;
  }
  if (count > h->u8_cap) {
    if (h->d_u8) HIPCHK(h, hipFree(h->d_u8));
    HIPCHK(h, hipMalloc((void**)&h->d_u8, count));
    h->u8_cap = count;
  }
  HIPCHK(h, launch_to_uint8(os.d, h->d_u8, (long long)count, st));
  HIPCHK(h, hipMemcpyAsync(out, h->d_u8, count, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  h->last_pending = false;
  return 0;
}

int ian_photo_blend(ian_handle* h, const float* z, const uint8_t* recon, const float* error, const double* gauss_half,
                    int32_t radius, uint8_t* im, double* mask, void* stream) {
  int rc = check_ready(h, 1);
  if (rc) return rc;
  if (!z || !recon || !error || !gauss_half || !im) return fail(h, -1, "null pointer passed to ian_photo_blend");
  if (radius < 0 || radius > 7) return fail(h, -7, "ian_photo_blend: radius %d outside 0..7", radius);
  Slot& os = h->slots[h->desc.out_slot];
  if (os.h != 64 || os.w != 64 || os.c != 3) return fail(h, -7, "ian_photo_blend needs a 3x64x64 image");
  hipStream_t st = (hipStream_t)stream;
  const bool graph = edit_graph_eligible(h, stream, {z, recon, error, im, mask});
  if (graph) {
    if ((rc = edit_ctx(h))) return rc;
    st = h->edit_stream;
  }
  enter_stream(h, st);
  TotalTimer tt(h, st);
  if (graph) {
    if ((rc = decode_one_graph(h, z))) return rc;
  } else if ((rc = decode_one_cached(h, z, st))) return rc;
  bool pending = false;
  if ((rc = enqueue_photo_blend(h, recon, error, gauss_half, radius, im, mask, st, &pending))) return rc;
  if (pending) {
    HIPCHK(h, hipStreamSynchronize(st));
    h->last_pending = false;
  }
  return 0;
}

int ian_brush_step(ian_handle* h, int32_t c1, int32_t r1, int32_t c2, int32_t r2, const float* rgb, const float* z, float coef,
                   float gscale, float* z_new, float* dz, float* x, const ian_photo_args* photo, void* stream) {
  int rc = check_ready(h, 1);
  if (rc) return rc;
  if (!z || !z_new) return fail(h, -1, "null pointer passed to ian_brush_step");
  for (const void* p : {(const void*)z, (const void*)z_new, (const void*)dz, (const void*)x})
    if (p && is_device_ptr(p)) return fail(h, -7, "ian_brush_step takes host buffers for z, z_new, dz, x (the Tk callback's arrays)");
  const int mode = rgb ? 1 : 0;
  const int zl = h->desc.num_latents;
  Slot& zs = h->slots[h->desc.z_slot];
  Slot& out = h->slots[h->desc.out_slot];
  if (c1 < 0 || r1 < 0 || c2 > out.w || r2 > out.h) return fail(h, -7, "patch (%d,%d,%d,%d) outside the %dx%d image", c1, r1, c2, r2, out.w, out.h);
  const bool photo_host = photo && !is_device_ptr(photo->recon) && !is_device_ptr(photo->error) && !is_device_ptr(photo->im) &&
                          !(photo->mask && is_device_ptr(photo->mask));
  if (!edit_graph_eligible(h, stream, {rgb}) || (photo && !photo_host)) {
    // the same step composed from the public calls (profiling on, a caller stream, graphs switched off, ...)
    std::vector<float> g((size_t)std::max(zs.c, zl)), zn((size_t)zl);
    if ((rc = grad_common(h, mode, c1, r1, c2, r2, rgb, z, g.data(), stream))) return rc;
    for (int i = 0; i < zl; ++i) {
      volatile float t = g[i] * gscale;   // volatile: one rounding per operation, as numpy does it
      volatile float u = coef * t;
      zn[i] = z[i] + u;
    }
    if (dz) memcpy(dz, g.data(), zl * sizeof(float));
    if (x) {
      if ((rc = ian_decode(h, zn.data(), 1, x, stream))) return rc;
    } else if (!photo) {
      hipStream_t st = (hipStream_t)stream;
      enter_stream(h, st);
      if ((rc = decode_one_cached(h, zn.data(), st))) return rc;
    }
    if (photo && (rc = ian_photo_blend(h, zn.data(), photo->recon, photo->error, photo->gauss_half, photo->radius, photo->im, photo->mask, stream)))
      return rc;
    memcpy(z_new, zn.data(), zl * sizeof(float));
    return 0;
  }
  if ((rc = edit_ctx(h))) return rc;
  hipStream_t st = h->edit_stream;
  enter_stream(h, st);
  if ((rc = decode_one_graph(h, z))) return rc;   // no-op when the resident activations already belong to z
  const float* d_rgb = nullptr;
  if (mode == 1 && (rc = upload_rgb_if_changed(h, rgb, st, &d_rgb))) return rc;
  if ((rc = ensure_slot(h, h->desc.z_slot, 1, true))) return rc;
  int* patch = reinterpret_cast<int*>(h->pin + PIN_IMG + 3 * 64 * 64);
  patch[0] = c1; patch[1] = r1; patch[2] = c2; patch[3] = r2;
  memcpy(patch + 4, &coef, sizeof(float));
  memcpy(patch + 5, &gscale, sizeof(float));
  h->dec_cache_valid = false;
  const int wx = x ? 1 : 0;
  rc = run_or_replay(h, h->g_step[mode][wx], [&]() -> int {
    HIPCHK(h, hipMemcpyAsync(h->d_patch, patch, 6 * sizeof(int), hipMemcpyHostToDevice, st));
    int r = run_decoder_backward(h, mode, 0, 0, out.w, out.h, d_rgb, st, h->d_patch);
    if (r) return r;
    HIPCHK(h, hipMemcpyAsync(h->pin + PIN_DZ, zs.g, zs.c * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(h, launch_latent_update(zs.d, zs.g, reinterpret_cast<const float*>(h->d_patch + 4), zl, st));
    HIPCHK(h, hipMemcpyAsync(h->pin + PIN_Z, zs.d, zl * sizeof(float), hipMemcpyDeviceToHost, st));
    if ((r = run_segment(h, IAN_SEG_DEC, 1, st))) return r;
    if (wx) HIPCHK(h, hipMemcpyAsync(h->pin + PIN_IMG, out.d, out.per_image() * sizeof(float), hipMemcpyDeviceToHost, st));
    return 0;
  });
  if (rc) return rc;
  bool pending = false;
  if (photo && (rc = enqueue_photo_blend(h, photo->recon, photo->error, photo->gauss_half, photo->radius, photo->im, photo->mask, st, &pending)))
    return rc;
  HIPCHK(h, hipStreamSynchronize(st));
  h->last_pending = false;
  memcpy(z_new, h->pin + PIN_Z, zl * sizeof(float));
  if (dz) memcpy(dz, h->pin + PIN_DZ, zl * sizeof(float));
  if (x) memcpy(x, h->pin + PIN_IMG, out.per_image() * sizeof(float));
  h->dec_cache_z.assign(z_new, z_new + zl);
  h->dec_cache_valid = true;
  return 0;
}

int ian_autotune(ian_handle* h, int32_t n, int32_t what, void* stream) {
  int rc = check_ready(h, n);
  if (rc) return rc;
  h->dec_cache_valid = false;
  hipStream_t st = (hipStream_t)stream;
  enter_stream(h, st);
  ++h->alloc_epoch;   // schedules are rebuilt below
  const bool prof = h->prof;
  h->prof = false;
  const bool verbose = getenv("IAN_DEBUG") != nullptr;
  static const char* cfg_names[TG_NCONFIG] = {"128x128", "128x64", "64x64", "32x128", "256x128", "128x32", "128x128/8w", "128x64/8w"};
  TuneCache cache;
  tune_cache_load(cache);
  bool cache_dirty = false;
  auto have = [&](int slot, bool grad) {
    if (slot < 0) return true;
    const Slot& s = h->slots[slot];
    return (grad ? s.gcap : s.cap) >= s.per_image() * (size_t)n && (grad ? s.g : s.d) != nullptr;
  };
  if (what & 1) {
    for (auto& op : h->ops) {
      if (!op.fwd.valid) continue;
      if (op.d.kind == IAN_OP_MDC3 && (mdc_head_eligible(h, op) || mdc_thin_eligible(h, op.fwd))) continue;  // VALU kernels: nothing to tune
      if (!have(op.d.src, false) || !have(op.d.src2, false) || !have(op.d.dst, false)) {
        h->prof = prof;
        return fail(h, -6, "ian_autotune: run a forward call with batch >= %d first (op '%s' has no activations)", n, op.name.c_str());
      }
      TgChoice best;
      float ms = 0;
      OpPlan* opp = &op;
      auto hit = cache.find(tune_key(n, "fwd", op.name));
      if (hit != cache.end()) {
        op.fwd.choice[n] = hit->second;
        free_schedule_for(op.fwd, n);
        continue;
      }
      cache_dirty = true;
      if ((rc = tune_layer(h, op.fwd, n, st, [&]() { return run_op_fwd(h, *opp, n, st); }, &best, &ms))) break;
      cache[tune_key(n, "fwd", op.name)] = best;
      if (verbose)
        fprintf(stderr, "[ian_autotune] n=%d fwd %-14s -> tile %s, max K-steps/item %d, schedule %d : %.1f us (%.1f TF/s)\n", n,
                op.name.c_str(), cfg_names[best.cfg], best.max_steps, best.variant, ms * 1e3,
                2.0 * op.fwd.macs_per_image() * n / (ms * 1e-3) / 1e12);
    }
  }
  if (!rc && (what & 2)) {
    if (n != 1) {
      h->prof = prof;
      return fail(h, -7, "ian_autotune: the latent-brush backward is a batch-1 path");
    }
    Slot& out = h->slots[h->desc.out_slot];
    if (!have(h->desc.out_slot, false)) rc = fail(h, -6, "ian_autotune: run ian_grad_* once first");
    for (auto& op : h->ops) {
      if (rc) break;
      if (op.d.segment != IAN_SEG_DEC || !op.bwd.valid || op.edge) continue;
      if (!have(op.d.dst, true)) {
        rc = fail(h, -6, "ian_autotune: run ian_grad_* once first");
        break;
      }
      TgChoice best;
      float ms = 0;
      auto hit = cache.find(tune_key(1, "bwd", op.name));
      if (hit != cache.end()) {
        op.bwd.choice[1] = hit->second;
        free_schedule_for(op.bwd, 1);
        continue;
      }
      cache_dirty = true;
      rc = tune_layer(h, op.bwd, 1, st, [&]() { return run_decoder_backward(h, 0, out.w / 2 - 2, out.h / 2 - 2, out.w / 2 + 2, out.h / 2 + 2, nullptr, st); }, &best, &ms);
      if (!rc) cache[tune_key(1, "bwd", op.name)] = best;
      if (!rc && verbose)
        fprintf(stderr, "[ian_autotune] n=1 bwd %-14s -> tile %s, max K-steps/item %d : chain %.1f us\n", op.name.c_str(),
                cfg_names[best.cfg], best.max_steps, ms * 1e3);
    }
  }
  h->prof = prof;
  if (!rc && cache_dirty) tune_cache_store(cache);
  return rc;
}

int ian_grad_rgb(ian_handle* h, int32_t c1, int32_t r1, int32_t c2, int32_t r2, const float* rgb, const float* z,
                 float* dz, void* stream) {
  if (!rgb) return fail(h, -1, "rgb is null");
  return grad_common(h, 1, c1, r1, c2, r2, rgb, z, dz, stream);
}

int ian_grad_light(ian_handle* h, int32_t c1, int32_t r1, int32_t c2, int32_t r2, const float* z, float* dz,
                   void* stream) {
  return grad_common(h, 0, c1, r1, c2, r2, nullptr, z, dz, stream);
}

static int read_slot_impl(ian_handle* h, int32_t slot, int32_t n, float* out, void* stream, bool grad) {
  int rc = check_ready(h, n);
  if (rc) return rc;
  if (slot < 0 || slot >= (int)h->slots.size()) return fail(h, -1, "bad slot %d", slot);
  hipStream_t st = (hipStream_t)stream;
  enter_stream(h, st);
  Slot s = h->slots[slot];
  if (grad) {
    s.d = s.g;
    s.cap = s.gcap;
    s.nchw = false;
  }
  if (!s.d || s.cap < s.per_image() * (size_t)n) return fail(h, -1, "slot %d holds no %s for batch %d", slot, grad ? "gradient" : "activation", n);
  if (!grad && h->slot_stale.count(slot) && h->slot_stale[slot])
    return fail(h, -1, "slot %d was bound to the caller's device buffer in the last call: the library holds no copy of it", slot);
  const size_t count = (size_t)n * s.c * s.h * s.w;
  const bool dev = is_device_ptr(out);
  float* tmp = nullptr;
  float* target = out;
  if (!dev) {
    HIPCHK(h, hipMalloc((void**)&tmp, count * sizeof(float)));
    target = tmp;
  }
  if (s.nchw) HIPCHK(h, hipMemcpyAsync(target, s.d, count * sizeof(float), hipMemcpyDeviceToDevice, st));
  else HIPCHK(h, launch_nhwc_to_nchw(s.d, s.cs, target, n, s.h * s.w, s.c, st));
  if (!dev) {
    HIPCHK(h, hipMemcpyAsync(out, tmp, count * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipFree(tmp));
  }
  return 0;
}

int ian_read_slot(ian_handle* h, int32_t slot, int32_t n, float* out, void* stream) {
  return read_slot_impl(h, slot, n, out, stream, false);
}

int ian_read_slot_grad(ian_handle* h, int32_t slot, int32_t n, float* out, void* stream) {
  return read_slot_impl(h, slot, n, out, stream, true);
}

int ian_profile_enable(ian_handle* h, int32_t on) {
  if (!h) return -1;
  h->prof = on != 0;
  h->ev_used = 0;
  h->ev_total_used = 0;
  h->prof_flops = 0;
  h->prof_launches = 0;
  return 0;
}

int ian_profile_read(ian_handle* h, double* tapgemm_ms, int64_t* tapgemm_launches, double* tapgemm_flops,
                     double* total_ms) {
  if (!h) return -1;
  HIPCHK(h, hipDeviceSynchronize());
  double ms = 0;
  for (size_t i = 0; i < h->ev_used; ++i) {
    float t = 0;
    HIPCHK(h, hipEventElapsedTime(&t, h->ev_pool[i].first, h->ev_pool[i].second));
    ms += t;
  }
  double tot = 0;
  for (size_t i = 0; i < h->ev_total_used; ++i) {
    float t = 0;
    HIPCHK(h, hipEventElapsedTime(&t, h->ev_total[i].first, h->ev_total[i].second));
    tot += t;
  }
  if (tapgemm_ms) *tapgemm_ms = ms;
  if (tapgemm_launches) *tapgemm_launches = h->prof_launches;
  if (tapgemm_flops) *tapgemm_flops = h->prof_flops;
  if (total_ms) *total_ms = tot;
  return 0;
}

int ian_set_option(ian_handle* h, const char* key, int32_t value) {
  if (!h || !key) return -1;
  if (!apply_option(h->opt, key, value)) return fail(h, -1, "unknown option '%s'", key);
  for (auto& op : h->ops) {  // schedules depend on the options
    free_schedules(op.fwd);
    free_schedules(op.bwd);
  }
  ++h->alloc_epoch;
  h->dec_cache_valid = false;   // the resident activations were produced under the previous options
  return 0;
}

void ian_destroy(ian_handle* h) {
  if (!h) return;
  IAN_GUARD_CHECK("ian_destroy");
  for (auto& op : h->ops) {
    free_schedules(op.fwd);
    free_schedules(op.bwd);
    for (TgLayer* L : {&op.fwd, &op.bwd}) {
      if (L->d_w) (void)hipFree(L->d_w);
      if (L->d_classes) (void)hipFree(L->d_classes);
      if (L->d_taps) (void)hipFree(L->d_taps);
    }
    for (float* p : {op.d_edge_w, op.d_scale, op.d_shift, op.d_made_w, op.d_made_b, op.d_b1_fwd, op.d_b1_bwd})
      if (p) (void)hipFree(p);
  }
  for (auto& s : h->slots) {
    if (s.d) (void)hipFree(s.d);
    if (s.g) (void)hipFree(s.g);
  }
  for (float* p : {h->d_slab, h->d_stage_in, h->d_stage_out, h->d_gseed, h->d_rgb, h->head.d_ftab, h->head.d_comp})
    if (p) (void)hipFree(p);
  if (h->head.d_itab) (void)hipFree(h->head.d_itab);
  for (ian_handle::EditGraph* g : {&h->g_fwd, &h->g_bwd[0], &h->g_bwd[1], &h->g_step[0][0], &h->g_step[0][1], &h->g_step[1][0], &h->g_step[1][1]})
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (h->edit_stream) (void)hipStreamDestroy(h->edit_stream);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->d_patch) (void)hipFree(h->d_patch);
  for (void* p : {(void*)h->d_recon, (void*)h->d_error, (void*)h->d_im, (void*)h->d_mask, (void*)h->d_u8})
    if (p) (void)hipFree(p);
  for (auto& e : h->ev_pool) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  for (auto& e : h->ev_total) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  delete h;
}

}  // extern "C"

// ====================================== layer objects (include/ian_train.h) ==============================
// One object per linear Lasagne layer of the training graph (train_IAN.py:116-149).  It reuses the inference
// packers to learn WHERE every reference-layout weight lands in the forward / backward slabs (the packers are run
// once on an index tensor), so that the per-step repacking after each optimiser update is a device-side gather.
struct WgSchedule {
  int cfg = 0, nitems = 0, nsplit = 1;
  WgItem* d_items = nullptr;
};

struct ian_layer {
  ian_handle ctx;  // options, split-K workspace, error text
  OpPlan op;
  bool is_mdc = false;
  std::vector<int64_t> pnumel;
  int* d_fwd_map = nullptr;
  int* d_bwd_map = nullptr;
  int* d_inv_map = nullptr;  // reference index -> forward slab index
  size_t fwd_count = 0, bwd_count = 0;
  MdcPackArgs mdc;
  int mdc_param_branch[1 + IAN_MAX_SCALES];  // parameter i (>=1) -> branch id, or -1 for the 1x1 coefficient
  float* d_partial = nullptr;
  size_t partial_cap = 0;
  float* d_dS = nullptr;  // MDC: reduced slab gradient
  // contract-first backward-weight of three sibling head layers (ian_layer_head6_backward_weight): owned by the first one
  ian_layer* h6_helper = nullptr;   // dense 128 -> 6*taps layer whose backward-weight is the one GEMM
  float* h6_Z = nullptr;
  size_t h6_Z_cap = 0;
  float* h6_dW = nullptr;
  int* h6_taps = nullptr;
  std::map<int, WgSchedule> wsched;
};

namespace {

int lfail(ian_layer* l, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (l) l->ctx.err = buf;
  return code;
}

#define LHIP(l, expr)                                                                                          \
  do {                                                                                                         \
    hipError_t _e = (expr);                                                                                    \
    if (_e != hipSuccess) return lfail(l, -2, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
  } while (0)

// index tensor: value = reference index + 1 (exact in float32 below 2^24), 0 marks padding
void add_index_param(ian_handle* h, const std::string& name, std::vector<int64_t> shape) {
  HostTensor t;
  t.shape = shape;
  const int64_t n = t.numel();
  t.data.resize(n);
  for (int64_t i = 0; i < n; ++i) t.data[i] = (float)(i + 1);
  h->params[name] = std::move(t);
}

std::vector<int> slab_to_map(const std::vector<float>& hw) {
  std::vector<int> m(hw.size());
  for (size_t i = 0; i < hw.size(); ++i) m[i] = (int)hw[i] - 1;
  return m;
}

int build_wg_schedule(ian_layer* l, int nimg, WgSchedule** out) {
  auto it = l->wsched.find(nimg);
  if (it != l->wsched.end()) {
    *out = &it->second;
    return 0;
  }
  const TgLayer& L = l->op.fwd;
  WgSchedule S;
  const int M = nimg * L.QH * L.QW;
  int bm, bn;
  if (L.Cout <= 32) { S.cfg = WG_32x128; bm = 32; bn = 128; }
  else if (L.Cin <= 32) { S.cfg = WG_128x32; bm = 128; bn = 32; }
  else { S.cfg = l->ctx.opt.wg_w8 ? WG_128x128W8 : WG_128x128; bm = 128; bn = 128; }
  const int tiles_co = (std::min(L.CoutPad, round_up(L.Cout, bm)) + bm - 1) / bm;
  const int tiles_ci = (L.Cin + bn - 1) / bn;
  const int ntaps = (int)L.taps.size();
  const long long base = (long long)ntaps * tiles_co * tiles_ci;
  const int steps = (M + 31) / 32;
  const int target = std::max(1, l->ctx.opt.wg_target_items);
  int nsplit = (int)std::max<long long>(1, (target + base - 1) / base);
  nsplit = std::min(nsplit, std::max(1, steps / 4));  // at least 4 K-steps (128 pixels) per item
  const int per = ((steps + nsplit - 1) / nsplit) * 32;
  nsplit = (M + per - 1) / per;
  S.nsplit = nsplit;
  std::vector<WgItem> items;
  for (int s = 0; s < nsplit; ++s)
    for (int c = 0; c < (int)L.classes.size(); ++c)
      for (int t = 0; t < L.classes[c].ntaps; ++t)
        for (int a = 0; a < tiles_co; ++a)
          for (int b = 0; b < tiles_ci; ++b) {
            WgItem wi{c, L.classes[c].tap0 + t, a * bm, b * bn, s * per, std::min(M, (s + 1) * per), s, 0};
            items.push_back(wi);
          }
  S.nitems = (int)items.size();
  int rc = upload(&l->ctx, items, &S.d_items);
  if (rc) return rc;
  const size_t slab_total = (size_t)ntaps * L.CoutPad * L.Cin;
  const size_t need = slab_total * nsplit;
  if (need > l->partial_cap) {
    if (l->d_partial) LHIP(l, hipFree(l->d_partial));
    LHIP(l, hipMalloc((void**)&l->d_partial, need * sizeof(float)));
    LHIP(l, hipMemset(l->d_partial, 0, need * sizeof(float)));
    l->partial_cap = need;
  }
  it = l->wsched.emplace(nimg, S).first;
  *out = &it->second;
  return 0;
}

}  // namespace

extern "C" {

const char* ian_layer_last_error(ian_layer* l) { return l ? l->ctx.err.c_str() : "null layer"; }

int ian_layer_create(const ian_op_desc* desc, int32_t deconv_flip, ian_layer** out) {
  if (!desc || !out) return -1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return -10;  // no HIP device: libian has no CPU fallback
  }
  std::unique_ptr<ian_layer> l(new ian_layer());
  ian_handle* h = &l->ctx;
  apply_env_options(h->opt);
  h->desc.deconv_flip = deconv_flip;
  h->finalized = true;
  OpPlan& op = l->op;
  op.d = *desc;
  op.name = "L";
  op.d.name = nullptr;
  op.d.bn_name = nullptr;
  op.d.has_bias = 0;
  const int cin = op.d.cin, cout = op.d.cout;
  int rc = 0;
  switch (op.d.kind) {
    case IAN_OP_CONV5S2:
      add_index_param(h, "L.W", {cout, cin, 5, 5});
      if ((rc = pack_conv_fwd(h, op, false)) || (rc = pack_conv_bwd(h, op))) return rc;
      l->pnumel = {(int64_t)cout * cin * 25};
      break;
    case IAN_OP_DECONV5S2:
      if (cout <= 4) return -4;  // image-producing deconvs are not part of the trainable graph (IAN.py:139-181)
      add_index_param(h, "L.W", {cin, cout, 5, 5});
      if ((rc = pack_deconv(h, op))) return rc;
      l->pnumel = {(int64_t)cout * cin * 25};
      break;
    case IAN_OP_DENSE: {
      add_index_param(h, "L.W", {cin, cout});
      std::vector<int> perm;
      bool has = false;
      if ((rc = pack_dense(h, op, perm, has))) return rc;
      l->pnumel = {(int64_t)cout * cin};
      break;
    }
    case IAN_OP_MDC3: {
      if (op.d.n_scales < 0 || op.d.n_scales > IAN_MAX_SCALES) return -1;
      // geometry / buffer sizes from the inference packer (values are overwritten by ian_layer_set_params)
      HostTensor w; w.shape = {cout, cin, 3, 3}; w.data.assign((size_t)cout * cin * 9, 0.f);
      h->params["LW"] = w;
      HostTensor c; c.shape = {cout}; c.data.assign(cout, 0.f);
      h->params["L_coeff_base"] = c;
      for (int i = 0; i < op.d.n_scales; ++i)
        h->params[op.d.scales[i] == 0 ? std::string("L_coeff_1x1") : "L_coeff_" + std::to_string(op.d.scales[i])] = c;
      if ((rc = pack_mdc(h, op))) return rc;
      l->is_mdc = true;
      const MdcTable T = mdc_table(op.d);
      if (T.taps.size() != op.fwd.taps.size() || T.taps.size() > 47) return -4;
      MdcPackArgs& a = l->mdc;
      memset(&a, 0, sizeof a);
      a.ntaps = (int)T.taps.size(); a.cout = cout; a.cin = cin; a.nbranch = (int)T.dil.size();
      a.f_rows = op.fwd.CoutPad; a.f_cols = op.fwd.Cin; a.b_rows = op.bwd.CoutPad; a.b_cols = op.bwd.Cin;
      int e = 0;
      for (int t = 0; t < a.ntaps; ++t) {
        a.tap_start[t] = e;
        for (int b = 0; b < a.nbranch; ++b)
          for (int pq = 0; pq < 9; ++pq)
            if (T.tapof[b][pq] == t) {
              if (e >= 48) return -4;
              a.ent_branch[e] = (unsigned char)b;
              a.ent_pq[e] = (unsigned char)pq;
              ++e;
            }
      }
      a.tap_start[a.ntaps] = e;
      l->pnumel = {(int64_t)cout * cin * 9, cout};
      l->mdc_param_branch[0] = 0;
      int nb = 1;
      for (int i = 0; i < op.d.n_scales; ++i) {
        l->pnumel.push_back(cout);
        l->mdc_param_branch[1 + i] = (op.d.scales[i] == 0) ? -1 : nb++;
      }
      break;
    }
    default:
      return -5;
  }
  if (!l->is_mdc) {
    std::vector<int> fm = slab_to_map(op.fwd.h_w), bm;
    if (op.bwd.valid) bm = slab_to_map(op.bwd.h_w);
    std::vector<int> inv((size_t)l->pnumel[0], -1);
    for (size_t i = 0; i < fm.size(); ++i)
      if (fm[i] >= 0) inv[fm[i]] = (int)i;
    l->fwd_count = fm.size();
    l->bwd_count = bm.size();
    if ((rc = upload(h, fm, &l->d_fwd_map)) || (rc = upload(h, bm, &l->d_bwd_map)) || (rc = upload(h, inv, &l->d_inv_map))) return rc;
  }
  if ((rc = upload_layer(h, op.fwd)) || (rc = upload_layer(h, op.bwd))) return rc;
  if (l->is_mdc) {
    l->mdc.slab_f = op.fwd.d_w;
    l->mdc.slab_b = op.bwd.d_w;
    if (hipMalloc((void**)&l->d_dS, op.fwd.w_floats * sizeof(float)) != hipSuccess) return -2;
  }
  h->params.clear();
  *out = l.release();
  return 0;
}

int32_t ian_layer_num_params(ian_layer* l) { return l ? (int32_t)l->pnumel.size() : -1; }
int64_t ian_layer_param_numel(ian_layer* l, int32_t which) {
  return (l && which >= 0 && which < (int)l->pnumel.size()) ? l->pnumel[which] : -1;
}

int ian_layer_set_params(ian_layer* l, const float* const* params, int32_t nparams, void* stream) {
  if (!l || !params || nparams != (int)l->pnumel.size()) return lfail(l, -1, "ian_layer_set_params: expected %d parameter tensors", l ? (int)l->pnumel.size() : 0);
  hipStream_t st = (hipStream_t)stream;
  if (!l->is_mdc) {
    LHIP(l, launch_gather_pack(params[0], l->d_fwd_map, l->op.fwd.d_w, (long long)l->fwd_count, st));
    if (l->op.bwd.valid) LHIP(l, launch_gather_pack(params[0], l->d_bwd_map, l->op.bwd.d_w, (long long)l->bwd_count, st));
    return 0;
  }
  MdcPackArgs& a = l->mdc;
  a.W = params[0];
  a.coeff_1x1 = nullptr;
  for (int i = 1; i < nparams; ++i) {
    const int b = (i == 1) ? 0 : l->mdc_param_branch[i - 1];
    if (i == 1) a.coeff[0] = params[1];
    else if (b < 0) a.coeff_1x1 = params[i];
    else a.coeff[b] = params[i];
  }
  LHIP(l, launch_mdc_pack(a, st));
  return 0;
}

int ian_layer_forward(ian_layer* l, const float* x, int32_t n, float* y, int32_t y_stride, const float* bias,
                      const float* res, int32_t act, void* stream) {
  if (!l || !x || !y || n <= 0) return lfail(l, -1, "bad argument to ian_layer_forward");
  TgEpilogue e;
  e.scale = nullptr; e.shift = bias; e.res = res; e.yfwd = nullptr; e.act = act; e.scale_period = 0; e.mode = TG_EPI_FWD;
  if (y_stride <= 0) y_stride = round_up(l->op.fwd.Cout, 32);
  if (l->op.d.kind == IAN_OP_MDC3 && mdc_head_eligible(&l->ctx, l->op)) {  // RGB-Beta head: VALU kernel
    const TgLayer& L = l->op.fwd;
    MdcHeadArgs a;
    memset(&a, 0, sizeof a);
    a.x = x; a.H = L.IH; a.W = L.IW; a.xs = L.Cin; a.ntaps = (int)L.taps.size();
    a.w_tap_stride = (long long)L.CoutPad * L.Cin;
    for (int t = 0; t < a.ntaps; ++t) { a.dy[t] = (signed char)L.taps[t].dy; a.dx[t] = (signed char)L.taps[t].dx; }
    if (bias) return lfail(l, -1, "head layers take no bias");
    for (int co = 0; co < L.Cout; ++co) {
      a.w[co] = L.d_w + (size_t)co * L.Cin; a.res[co] = res; a.y[co] = y; a.ys[co] = y_stride; a.yc[co] = co; a.act[co] = act;
      a.scale[co] = 1.f; a.shift[co] = 0.f;
    }
    LHIP(l, launch_mdc_head(a, n, L.Cin, L.Cout, (hipStream_t)stream));
    return 0;
  }
  if (l->op.d.kind == IAN_OP_MDC3 && mdc_thin_eligible(&l->ctx, l->op.fwd)) {
    MdcThinArgs a;
    mdc_thin_fill(a, l->op.fwd, x, l->op.fwd.Cin, y, y_stride, n, &l->ctx);
    a.res = res; a.shift = bias; a.act = act;
    LHIP(l, launch_mdc_thin(a, (hipStream_t)stream));
    return 0;
  }
  return run_tapgemm(&l->ctx, l->op.fwd, n, x, y, y_stride, e, (hipStream_t)stream);
}

int ian_layer_backward_data(ian_layer* l, const float* dy, int32_t n, float* dx, int32_t dx_stride, int32_t accumulate,
                            void* stream) {
  if (!l || !dy || !dx || n <= 0) return lfail(l, -1, "bad argument to ian_layer_backward_data");
  if (!l->op.bwd.valid) return lfail(l, -9, "layer has no backward-data form");
  TgEpilogue e;
  e.scale = nullptr; e.shift = nullptr; e.res = accumulate ? dx : nullptr; e.yfwd = nullptr; e.act = IAN_ACT_NONE;
  e.scale_period = 0; e.mode = TG_EPI_BWD;
  if (dx_stride <= 0) dx_stride = round_up(l->op.bwd.Cout, 32);
  if (l->op.d.kind == IAN_OP_MDC3 && mdc_thin_eligible(&l->ctx, l->op.bwd)) {
    MdcThinArgs a;
    mdc_thin_fill(a, l->op.bwd, dy, l->op.bwd.Cin, dx, dx_stride, n, &l->ctx);
    a.res = accumulate ? dx : nullptr;
    LHIP(l, launch_mdc_thin(a, (hipStream_t)stream));
    return 0;
  }
  return run_tapgemm(&l->ctx, l->op.bwd, n, dy, dx, dx_stride, e, (hipStream_t)stream);
}

int ian_layer_backward_weight(ian_layer* l, const float* x, const float* dy, int32_t n, float* const* dparams,
                              int32_t nparams, int32_t accumulate, void* stream) {
  if (!l || !x || !dy || !dparams || n <= 0 || nparams != (int)l->pnumel.size()) return lfail(l, -1, "bad argument to ian_layer_backward_weight");
  hipStream_t st = (hipStream_t)stream;
  const TgLayer& L = l->op.fwd;
  const bool thin_in = l->is_mdc && l->ctx.opt.mdc_head && L.cin_real <= 4 && L.Cout <= 4 && (L.IH % 4) == 0 && (L.IW % 16) == 0 &&
                       L.taps.size() <= 48 && mdc_thin_eligible(&l->ctx, L);
  if (l->is_mdc && (mdc_head_eligible(&l->ctx, l->op) || thin_in)) {  // RGB-Beta head: VALU backward-weight, no padded MFMA tile
    const int kcin = thin_in ? 4 : L.Cin;  // thin layers: the 2-4 real channels (+ zero padding) of the 32-float pixel row
    MdcHeadWgradArgs a;
    memset(&a, 0, sizeof a);
    a.x = x; a.dy = dy; a.H = L.IH; a.W = L.IW; a.xs = L.Cin; a.dys = round_up(L.Cout, 32); a.ntaps = (int)L.taps.size();
    a.total_tiles = n * (L.IH / 4) * (L.IW / 16);
    for (int t = 0; t < a.ntaps; ++t) { a.dy_[t] = (signed char)L.taps[t].dy; a.dx_[t] = (signed char)L.taps[t].dx; }
    const int nblocks = std::min(a.total_tiles, 512);
    const int cpad = L.Cout <= 2 ? 2 : 4;
    const size_t need = (size_t)nblocks * a.ntaps * cpad * kcin;
    if (need > l->partial_cap) {
      if (l->d_partial) LHIP(l, hipFree(l->d_partial));
      LHIP(l, hipMalloc((void**)&l->d_partial, need * sizeof(float)));
      l->partial_cap = need;
    }
    a.partial = l->d_partial;
    LHIP(l, hipMemsetAsync(l->d_dS, 0, l->op.fwd.w_floats * sizeof(float), st));
    LHIP(l, launch_mdc_head_wgrad(a, nblocks, kcin, L.Cout, l->d_dS, L.CoutPad, L.Cin, st));
    MdcCoeffGrads g;
    memset(&g, 0, sizeof g);
    g.d[0] = dparams[1];
    for (int i = 2; i < nparams; ++i) {
      const int b = l->mdc_param_branch[i - 1];
      if (b < 0) g.d1x1 = dparams[i];
      else g.d[b] = dparams[i];
    }
    LHIP(l, launch_mdc_unpack_grad(l->mdc, l->d_dS, dparams[0], g, accumulate, st));
    return 0;
  }
  WgSchedule* S;
  int rc = build_wg_schedule(l, n, &S);
  if (rc) return rc;
  WgParams p;
  p.x = x; p.dy = dy; p.partial = l->d_partial; p.items = S->d_items; p.classes = L.d_classes; p.taps = L.d_taps;
  p.M = n * L.QH * L.QW; p.IH = L.IH; p.IW = L.IW; p.Cin = L.Cin;
  p.qw_shift = ilog2_exact(L.QW); p.qhw_shift = ilog2_exact(L.QH * L.QW);
  p.si = L.si; p.by = L.by; p.bx = L.bx; p.so = L.so; p.OH = L.OH; p.OW = L.OW;
  p.dy_stride = (l->op.d.kind == IAN_OP_DENSE && l->op.d.unflat_c > 0) ? L.Cout : round_up(L.Cout, 32);
  p.CoutPad = L.CoutPad; p.CinPad = L.Cin;
  p.slab_total = (long long)L.taps.size() * L.CoutPad * L.Cin;
  const size_t xb = (size_t)n * L.IH * L.IW * L.Cin * 4, yb = (size_t)n * L.OH * L.OW * p.dy_stride * 4;
  if (xb > 0xFFFFFFF0ull || yb > 0xFFFFFFF0ull) return lfail(l, -7, "batch %d exceeds the 4 GiB buffer-descriptor range", n);
  p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
  LHIP(l, launch_tapwgrad(S->cfg, p, S->nitems, st));
  if (!l->is_mdc) {
    LHIP(l, launch_wgrad_reduce(l->d_partial, p.slab_total, S->nsplit, l->d_inv_map, dparams[0], l->pnumel[0], accumulate, st));
    return 0;
  }
  LHIP(l, launch_wgrad_reduce(l->d_partial, p.slab_total, S->nsplit, nullptr, l->d_dS, p.slab_total, 0, st));
  MdcCoeffGrads g;
  memset(&g, 0, sizeof g);
  g.d[0] = dparams[1];
  for (int i = 2; i < nparams; ++i) {
    const int b = l->mdc_param_branch[i - 1];
    if (b < 0) g.d1x1 = dparams[i];
    else g.d[b] = dparams[i];
  }
  LHIP(l, launch_mdc_unpack_grad(l->mdc, l->d_dS, dparams[0], g, accumulate, st));
  return 0;
}

int ian_layer_head6_forward(ian_layer* l0, ian_layer* l1, ian_layer* l2, const float* x, int32_t n, float* y0, float* y1,
                            float* y2, int32_t y_stride, int32_t act0, int32_t act1, int32_t act2, void* stream) {
  if (!l0 || !l1 || !l2 || !x || !y0 || !y1 || !y2 || n <= 0) return lfail(l0, -1, "bad argument to ian_layer_head6_forward");
  hipStream_t st = (hipStream_t)stream;
  ian_layer* ls[3] = {l0, l1, l2};
  const TgLayer& L = l0->op.fwd;
  for (ian_layer* l : ls) {
    const TgLayer& M = l->op.fwd;
    if (!l->is_mdc || l->op.d.cin != 128 || l->op.d.cout != 2 || l->op.d.in_w != 64 || M.Cin != 128 || !same_taps(L, M)) return -4;
  }
  if (L.taps.size() > 37) return -4;
  ian_handle* h = &l0->ctx;
  HeadPlan& P = h->head;   // tables + compact workspace live in the first layer's context
  if (!P.d_itab) {
    std::vector<int> itab(44, -1), used(9, 0);
    int halo = 0;
    for (size_t t = 0; t < L.taps.size(); ++t) {
      const int dy = L.taps[t].dy, dx = L.taps[t].dx;
      if (dy < -4 || dy > 4 || dx < -63 || dx > 63) return -4;
      const int g = dy == 0 ? 0 : (dy < 0 ? dy + 5 : dy + 4), cap = g == 0 ? 12 : 4, s0 = g == 0 ? 0 : 12 + (g - 1) * 4;
      if (used[g] >= cap) return -4;
      itab[s0 + used[g]++] = (int)t | ((dx + 64) << 8);
      halo = std::max(halo, std::abs(dy));
    }
    P.halo = halo;
    int rc = upload(h, itab, &P.d_itab);
    if (rc) return rc;
    LHIP(l0, hipMalloc((void**)&P.d_ftab, 32 * sizeof(float)));
  }
  const int acts[3] = {act0, act1, act2};
  const int act_key = 1 + act0 + 16 * act1 + 256 * act2;
  if (P.opBeta != act_key) {   // (field reused as "epilogue table uploaded for these activations"): synchronous, first call only
    float ftab[32] = {0};
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 2; ++c) { ftab[2 * k + c] = 1.f; ftab[16 + 2 * k + c] = (float)acts[k]; }
    LHIP(l0, hipStreamSynchronize(st));
    LHIP(l0, hipMemcpy(P.d_ftab, ftab, sizeof ftab, hipMemcpyHostToDevice));
    P.opBeta = act_key;
  }
  const int H = l0->op.d.in_h, W = l0->op.d.in_w;
  const size_t need = (size_t)n * H * W * 8;
  if (need > P.comp_cap) {
    if (P.d_comp) LHIP(l0, hipFree(P.d_comp));
    LHIP(l0, hipMalloc((void**)&P.d_comp, need * sizeof(float)));
    P.comp_cap = need;
  }
  HeadFusedArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.w0 = l0->op.fwd.d_w; a.w1 = l1->op.fwd.d_w; a.w2 = l2->op.fwd.d_w; a.out = P.d_comp; a.itab = P.d_itab; a.ftab = P.d_ftab;
  a.H = H; a.W = W; a.xs = L.Cin; a.ntaps = (int)L.taps.size(); a.halo = P.halo;
  a.w_tap_stride = (long long)L.CoutPad * L.Cin;
  int bands = 1;
  while (n * bands < 256 && bands < 8 && (H % (bands * 2)) == 0 && H / (bands * 2) >= 2 * P.halo) bands *= 2;
  a.bands = bands;
  LHIP(l0, launch_head6(a, n, st));
  LHIP(l0, launch_head6_scatter(P.d_comp, y0, y1, y2, y_stride, (long long)n * H * W, st));
  return 0;
}

int ian_layer_head6_backward(ian_layer* l0, ian_layer* l1, ian_layer* l2, const float* x, const float* dy0, const float* dy1,
                             const float* dy2, int32_t n, int32_t dy_stride, float* dx, int32_t dx_stride, int32_t dx_accumulate,
                             float* const* dparams0, float* const* dparams1, float* const* dparams2, int32_t nparams,
                             int32_t accumulate, void* stream) {
  const bool want_w = dparams0 || dparams1 || dparams2;
  if (!l0 || !l1 || !l2 || !dy0 || !dy1 || !dy2 || n <= 0 || (!dx && !want_w) || (want_w && (!x || !dparams0 || !dparams1 || !dparams2)))
    return lfail(l0, -1, "bad argument to ian_layer_head6_backward");
  hipStream_t st = (hipStream_t)stream;
  ian_layer* ls[3] = {l0, l1, l2};
  float* const* dps[3] = {dparams0, dparams1, dparams2};
  const TgLayer& L = l0->op.fwd;
  for (ian_layer* l : ls) {
    const TgLayer& M = l->op.fwd;
    if (!l->is_mdc || l->op.d.cin != 128 || l->op.d.cout != 2 || M.Cin != 128 || !same_taps(L, M) ||
        (want_w && nparams != (int)l->pnumel.size()))
      return -4;
  }
  const int ntaps = (int)L.taps.size(), nout = 6 * ntaps, zs = round_up(nout, 32);
  const int H = l0->op.d.in_h, W = l0->op.d.in_w;
  const long long npix = (long long)n * H * W;
  if (npix * zs * 4 > 0xFFFFFFF0ll || npix > 0x7FFFFFFFll) return -4;   // buffer-descriptor range of the helper GEMMs
  if (!l0->h6_helper) {
    ian_op_desc d;
    memset(&d, 0, sizeof d);
    d.kind = IAN_OP_DENSE; d.cin = 128; d.cout = nout; d.in_h = d.in_w = 1; d.src2 = d.src3 = -1;
    int rc = ian_layer_create(&d, 1, &l0->h6_helper);
    if (rc) return lfail(l0, rc, "head6 backward: helper layer creation failed (%d)", rc);
    std::vector<int> tp(ntaps);
    for (int t = 0; t < ntaps; ++t) tp[t] = (L.taps[t].dy + 64) | ((L.taps[t].dx + 64) << 8);
    if ((rc = upload(&l0->ctx, tp, &l0->h6_taps))) return rc;
    LHIP(l0, hipMalloc((void**)&l0->h6_dW, (size_t)2 * 128 * nout * sizeof(float)));   // [0]: dW of the helper, [1]: its W
  }
  const size_t need = (size_t)npix * zs;
  if (need > l0->h6_Z_cap) {
    if (l0->h6_Z) LHIP(l0, hipFree(l0->h6_Z));
    LHIP(l0, hipMalloc((void**)&l0->h6_Z, need * sizeof(float)));
    l0->h6_Z_cap = need;
  }
  LHIP(l0, launch_head6_zbuild(dy0, dy1, dy2, dy_stride, l0->h6_Z, zs, H, W, ntaps, l0->h6_taps, npix, st));
  int rc;
  if (dx) {  // dx[q][c] = sum_j Z[q][j] * Wcat[c][j], Wcat[c][(t,k,f)] = slab_k[t][f][c]: the helper's backward-data GEMM
    float* wcat = l0->h6_dW + (size_t)128 * nout;
    for (int k = 0; k < 3; ++k)
      LHIP(l0, launch_head6_wcat(ls[k]->op.fwd.d_w, ls[k]->op.fwd.CoutPad, ls[k]->op.fwd.Cin, k, ntaps, wcat, nout, st));
    float* wp[1] = {wcat};
    if ((rc = ian_layer_set_params(l0->h6_helper, wp, 1, stream)) ||
        (rc = ian_layer_backward_data(l0->h6_helper, l0->h6_Z, (int32_t)npix, dx, dx_stride, dx_accumulate, stream)))
      return lfail(l0, rc, "head6 backward-data GEMM: %s", ian_layer_last_error(l0->h6_helper));
  }
  if (!want_w) return 0;
  float* dw_ptr[1] = {l0->h6_dW};
  rc = ian_layer_backward_weight(l0->h6_helper, x, l0->h6_Z, (int32_t)npix, dw_ptr, 1, 0, stream);
  if (rc) return lfail(l0, rc, "head6 backward-weight GEMM: %s", ian_layer_last_error(l0->h6_helper));
  for (int k = 0; k < 3; ++k) {
    ian_layer* l = ls[k];
    LHIP(l, hipMemsetAsync(l->d_dS, 0, l->op.fwd.w_floats * sizeof(float), st));
    LHIP(l, launch_head6_dS_scatter(l0->h6_dW, nout, k, ntaps, l->d_dS, l->op.fwd.CoutPad, l->op.fwd.Cin, st));
    MdcCoeffGrads g;
    memset(&g, 0, sizeof g);
    g.d[0] = dps[k][1];
    for (int i = 2; i < nparams; ++i) {
      const int b = l->mdc_param_branch[i - 1];
      if (b < 0) g.d1x1 = dps[k][i];
      else g.d[b] = dps[k][i];
    }
    LHIP(l, launch_mdc_unpack_grad(l->mdc, l->d_dS, dps[k][0], g, accumulate, st));
  }
  return 0;
}

int ian_layer_autotune(ian_layer* l, int32_t n, float* scratch_a, float* scratch_b, int64_t cap_floats, void* stream) {
  if (!l || !scratch_a || !scratch_b || n <= 0) return lfail(l, -1, "bad argument to ian_layer_autotune");
  hipStream_t st = (hipStream_t)stream;
  ian_handle* h = &l->ctx;
  TgEpilogue e;
  e.scale = nullptr; e.shift = nullptr; e.res = nullptr; e.yfwd = nullptr; e.act = IAN_ACT_NONE; e.scale_period = 0; e.mode = TG_EPI_FWD;
  struct Dir { TgLayer* L; float* in; float* out; };
  Dir dirs[2] = {{&l->op.fwd, scratch_a, scratch_b}, {&l->op.bwd, scratch_b, scratch_a}};
  for (int d = 0; d < 2; ++d) {
    TgLayer& L = *dirs[d].L;
    if (!L.valid) continue;
    if (d == 0 && l->op.d.kind == IAN_OP_MDC3 && (mdc_head_eligible(h, l->op) || mdc_thin_eligible(h, L))) continue;  // VALU kernels
    if (d == 1 && l->op.d.kind == IAN_OP_MDC3 && mdc_thin_eligible(h, L)) continue;
    const int ystride = round_up(L.Cout, 32);
    const int64_t need_in = (int64_t)n * L.IH * L.IW * L.Cin, need_out = (int64_t)n * L.OH * L.OW * ystride;
    if (need_in > cap_floats || need_out > cap_floats) return lfail(l, -7, "ian_layer_autotune: scratch buffers too small (%lld floats needed)", (long long)std::max(need_in, need_out));
    float* in = dirs[d].in;
    float* out = dirs[d].out;
    int rc = tune_layer(h, L, n, st, [&]() { return run_tapgemm(h, L, n, in, out, ystride, e, st); }, nullptr, nullptr);
    if (rc) return rc;
  }
  return 0;
}

void ian_layer_destroy(ian_layer* l) {
  IAN_GUARD_CHECK("ian_layer_destroy");
  if (!l) return;
  if (l->h6_helper) ian_layer_destroy(l->h6_helper);
  for (void* p : {(void*)l->h6_Z, (void*)l->h6_dW, (void*)l->h6_taps})
    if (p) (void)hipFree(p);
  for (TgLayer* L : {&l->op.fwd, &l->op.bwd}) {
    free_schedules(*L);
    if (L->d_w) (void)hipFree(L->d_w);
    if (L->d_classes) (void)hipFree(L->d_classes);
    if (L->d_taps) (void)hipFree(L->d_taps);
  }
  for (auto& kv : l->wsched)
    if (kv.second.d_items) (void)hipFree(kv.second.d_items);
  for (void* p : {(void*)l->d_fwd_map, (void*)l->d_bwd_map, (void*)l->d_inv_map, (void*)l->d_partial, (void*)l->d_dS,
                  (void*)l->ctx.d_slab, (void*)l->ctx.head.d_itab, (void*)l->ctx.head.d_ftab, (void*)l->ctx.head.d_comp})
    if (p) (void)hipFree(p);
  delete l;
}

}  // extern "C"
