// ian_rt_types.h -- types of the runtime: packed layers, slots, op plans, options, the handle.
// Part of the libian runtime: one translation unit, included by ian_runtime.cpp in this order (see the list there).
#pragma once
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
  float* d_raw = nullptr;      // fused == 2: one accumulation tile per output tile (zero at rest)
  int fused = 0;               // how this schedule's split-K partials are combined (TgParams::fused)
  bool bf16x3 = false;         // launches go to tapgemm_bf16x3_kernel (option tg_bf16x3 and a tile it is instantiated for)
  size_t slab_tiles = 0;
  int max_nsplit = 1;
  std::vector<TgItem> h_items;  // kept for tests / debugging
  std::vector<TgTile> h_tiles;
};

struct TgChoice {  // autotuned (or forced) schedule shape for one (layer, batch)
  int cfg = -1;        // enum TgConfig, -1 = heuristic
  int max_steps = -1;  // -1 = heuristic, 0 = never split K, >0 = split so that no item exceeds this many K-steps
  int variant = -1;    // K-loop schedule of tapgemm_kernel, -1 = the handle's option
  int fused = -1;      // split-K combine: -1 = the handle's option (tg_fuse for small M), 0 reduce launch, 1 in-launch slabs, 2 atomics
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
  unsigned short* d_wsplit = nullptr;   // d_w split into a bf16 hi plane + lo plane (2 x w_floats values) for tapgemm_bf16x3_kernel; made lazily
  bool wsplit_valid = false;            // ... and re-made after d_w changed (training layers repack their slabs after every update)
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
  int tg_xcd_spatial = 0;            // experiment (round 4, OFF): equal-weight runs of tile groups dealt to the XCDs in contiguous pieces instead of
                                     // round-robin.  Measured: same step time, HBM fetch per tapgemm launch 85 -> 122 MB at batch 64
  int tg_prefer_nosplit = 1;  // try smaller tiles before resorting to split-K
  int tg_nosplit_min_out = 1 << 30;  // outputs (M*Cout) above which 64x64 is forced even if it under-fills
  int mdc_head = 2;                  // few-filter MDCL layers: 0 = tapgemm, 1 = VALU head kernel, 2 = + sibling layers fused
  int tg_variant = 2;                // K-loop schedule of tapgemm_kernel (kernels_tapgemm.hip); autotune picks per layer
  int tg_fuse = 0;                   // split-K combine of launches with images*QH*QW <= tg_fuse_max_m (the batch-1 chains, where a reduce launch
                                     // costs as much as the K loop): 0 = separate tapgemm_reduce launch; 1 = write-through (sc1) slabs summed by
                                     // the tile's last-arriving workgroup in slice order (deterministic); 2 = float atomics into a zero-at-rest
                                     // tile + epilogue by the last arriver (run-to-run last-bit differences).  The autotuner may pick per layer.
  int tg_fuse_max_m = 1024;
  int tg_fuse_tune = 0;              // fused modes ian_autotune may try for such launches: 0 none, 1 = mode 1, 2 = modes 1 and 2.  OFF: measured on
                                     // MI355X (profiles/r05_b1_inlaunch_combine_ab.json) the autotuner kept the reduce launch for every 5x5 layer, and
                                     // the fused modes forced on untuned layers cost +13 us (mode 1) / +9 us (mode 2) per layer
  int tg_bf16x3 = 0;                 // OPT-IN, never the default: tap-GEMMs with at least tg_bf16x3_min_m rows on the bf16 matrix cores, every fp32
                                     // product as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi of split-bf16 operands with fp32 accumulation (kernels_tapgemm.hip
                                     // tapgemm_bf16x3_kernel): ~1e-5 relative error instead of exact fp32 -- inside the 1e-4 parity bar, but a different
                                     // arithmetic type, so bench.py reports it as a labelled secondary only
  int tg_bf16x3_min_m = 256;
  int tg_bf16x3_min_images = 2;      // ... and never to batch-1 launches: the interactive chains are latency-bound (no use for matrix rate), and the
                                     // brush GRADIENT is discontinuous in the forward activations (ReLU / leaky-ReLU kinks): measured, a 1e-5 forward
                                     // error flips a unit under the patch and moves dL/dz by 1e-3 on 2 of the 6 fixture patches, while split-bf16
                                     // backward-data alone stays within 1.1e-5 (scripts/exp/bf16x3_brush_debug.py)
  int tg_bf16x3_wsplit = 1;          // weights pre-split once per layer (0: split at staging like the activations)
  int tg_bf16x3_sched = 1;           // K-loop schedule of tapgemm_bf16x3_kernel (0..2, kernels_tapgemm.hip)
  int tg_bf16x3_fwd = 1, tg_bf16x3_bwd = 1;   // which epilogue modes (forward / backward-data launches) the option applies to
  int tg_tune_deep = 1;              // 1: autotune also times K-loop schedule 7 (the rotated schedule with two K-steps of loads in flight)
  int tg_tune_model = 1;             // 1: autotune also times the split limits a longest-first model of the launch ranks best (ian_rt_autotune.inc)
  int tg_tune_final = 6;             // autotune: the N fastest candidates of the first pass (5 launches each) are timed again in 3 interleaved rounds of 12 launches; 0 = first pass decides
  int tg_variant_force = -1;         // >= 0: every tapgemm launch takes this K-loop schedule, also over an autotuned choice (A/B and ablation timing: scripts/exp/tg_ablate7.py)
  int tg_noepi = 0;                  // libian_ablation.so only: tapgemm returns right after its K loop (timing-only ablation, wrong results)
  int tg_fast_epilogue = 1;          // tapgemm: the per-row / per-column epilogue (tg_store_fast) where it applies; 0 = the general form everywhere (A/B, bitwise the same)
  int tg_tune_pin = 0;               // 1: autotune also times K-loop schedule 6 (schedule 2 with its fragment reads pinned; kernels_tapgemm.hip).  OFF: measured,
                                     // 44.28 / 44.03 k reconstructions/s without vs 44.13 / 44.01 k with the candidate: no gain, 25 % more tuning time
  int tg_reduce_kp = 4;              // split-K reduce: lanes sharing one output element's slabs when a tile has >= 8 slabs
  int wg_w8 = 1;                     // tapwgrad: 8-wave 128x128 workgroups (16 waves per CU instead of 8)
  int wg_reduce_tci = 16;            // tiled split reduce / repack: channels per tile row (16: 64-byte runs; 32: 128-byte runs, half the filters per tile)
  int wg_pipe = 3;                   // tapwgrad: the software-pipelined K loop of the 8-wave tile (tapwgrad_p_kernel; 0 = the compiler-scheduled loop, bitwise the same)
  int zbuild_rows = 1;               // head6 backward: Z built a pixel row at a time through LDS (head6_zbuild_rows_kernel; 0 = per (pixel, tap), same bytes)
  int pack_tiled = 1;                // training layers repack their slabs after an update through LDS tiles (pack_tiled_kernel; 0 = the gather, same bytes)
  int wg_reduce_tiled = 1;           // tapwgrad's split reduce scatters to the reference layout through LDS (wgrad_reduce_tiled_kernel; 0 = the gather, bitwise the same)
  int wg_xcd_split = 1;              // tapwgrad: all items of one pixel range go to one XCD (its rows are fetched into that L2 once), split count a
                                     // multiple of 8 (round 5); 0 = round 4's (split, tap) order
  int wg_target_items = 1024;        // tapwgrad: split the pixel range until taps x channel tiles x splits reaches this many workgroups
  int dec_out_bal = 1;               // dec_out (Cout = 3) on 16 x 16 MFMA blocks, five per SIMD (deconv_small_kernel<3, true>); 0 = the 32 x 32 tiling (six of eight waves)
  int mdc_thin_tile = 1;             // thin MDCL (G_b / B_b and their backward-data) with the input rows staged through LDS
  int b1_conv = 0;                   // batch-1 transposed convs and their backward-data as whole-contraction streaming launches
                                     // (kernels_b1.hip).  OFF: measured slower than tapgemm + reduce (brush event 0.178 vs 0.160 ms):
                                     // 16-pixel tiles re-read weights and input rows from L2 at 4 FLOP/B
  int dec_out_wgs = 256;             // image-producing deconv: split images into row bands until this many workgroups exist
  int alias_io = 1;                  // ian_reconstruct: device-pointer images are read / written in place (no boundary copies)
  int dense_gemv = 1;                // batch-1 backward of the dense layer fed by the latent as one GEMV launch
  int dec_out_px = 1;                // ... and, below that batch, 8 lanes per output pixel instead of 16 tile workgroups
  int dec_out_mfma = 1;              // image-producing deconv (IAN_simple dec_out) on the matrix cores for batches >= 4
  int fuse_latent_update = 1;        // ian_brush_step: the latent update rides in the epilogue of the latent's backward GEMV (round 5)
  int edit_graph = 1;                // batch-1 host-pointer calls (the NPE edit loop) replay captured hipGraphs
  int edit_spin = 1;                 // ... and their final wait polls the stream (hipStreamQuery) before it falls back to hipStreamSynchronize
  int edit_keep_warm_us = 0;         // EXPERIMENT (off): after a brush event a one-wave kernel keeps the interactive stream's queue busy for up to
                                     // this many microseconds, released by the next call (kernels_npe.hip keep_warm_kernel)
  int edit_zero_copy = 1;            // their kernels read the brush rectangle from / write z, dz, the image to the pinned block directly
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
  std::map<int, char> slot_aliased;  // ... and which of them are bound to a caller's buffer right now
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
    bool mirrors = false;        // its image kernel writes the pinned image as well (zero-copy): no device -> host copy needed
  };
  long long alloc_epoch = 0;
  hipStream_t edit_stream = nullptr;
  float* pin = nullptr;      // pinned host block: z [0,128) | dz [128,256) | image [256, 256+12288) | patch (4 ints) after that
  // Round 4: the block is allocated mapped + coherent and the kernels of the interactive graphs read the brush rectangle from it
  // and write the new latent, the gradient and the image into it directly (pin_dev = its device address; nullptr: not
  // available, the graphs keep their copy nodes).  Every hipMemcpyAsync of a graph is a 4.6 us copy kernel plus a kernel
  // boundary on this runtime: four of them per brush event.
  float* pin_dev = nullptr;
  // training step (ian_layer_stats_next): the NEXT tap-GEMM launch through this context also produces the batch statistics of
  // the tensor it stores (TgStats); one-shot.  stats_chunks_last = partial rows the last launch wrote (0: it could not -- split-K
  // schedule, VALU path, workspace too small -- and the caller runs colstats instead)
  TgStats stats_next;
  long long stats_cap = 0;       // doubles available at stats_next.partial
  int stats_chunks_last = 0;
  float* out_mirror = nullptr;   // set around run_segment(DEC): the batch-1 image kernel also writes the image there
  // set around run_decoder_backward by ian_brush_step: the latent's backward GEMV also applies Z += coef * (dZ * gscale) (one launch
  // less per brush event); done = the fused form was taken (else the caller launches latent_update_kernel)
  struct { const float* cg = nullptr; float* z_mirror = nullptr; float* g_mirror = nullptr; bool done = false; } upd;
  bool warm_armed = false;       // a keep_warm_kernel is (or may still be) spinning on edit_stream: enter_stream releases it
  bool pin_img_valid = false;    // pin[PIN_IMG..] holds the image that is resident in the output slot
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

