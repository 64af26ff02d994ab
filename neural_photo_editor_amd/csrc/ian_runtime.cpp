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
#include <functional>
#include <memory>
#include <queue>
#include <string>
#include <vector>

#include "../../include/ian.h"
#include "../../include/ian_train.h"
#include "ian_internal.h"
#include "ian_guard.h"   // IAN_SANITIZE builds: guard bands around every device allocation (no-op otherwise)

using namespace ian;


// The runtime is ONE translation unit split by concern (round 3; the file had grown to 3 000 lines).  Order matters: each
// part may use what the parts above it define.
#include "ian_rt_types.h"      // packed layers, slots, op plans, options, the handle
#include "ian_rt_util.inc"     // options, error text, parameter lookup, uploads
#include "ian_rt_pack.inc"     // epilogue vectors, weight repacking
#include "ian_rt_schedule.inc" // tapgemm item tables: tiles, split-K, XCD-aware order
#include "ian_rt_exec.inc"     // slots, launches, fused RGB-Beta head, forward executor
#include "ian_rt_autotune.inc" // per (layer, batch) schedule choice by timing + its cache
#include "ian_rt_io.inc"       // images / latents in and out, device-pointer aliasing
#include "ian_rt_backward.inc" // latent-brush backward sweep
#include "ian_rt_edit.inc"     // interactive loop: streams, captured graphs, decoder cache, photo blend
#include "ian_rt_api.inc"      // extern "C": include/ian.h
#include "ian_rt_layer.inc"    // extern "C": ian_layer_* of include/ian_train.h
