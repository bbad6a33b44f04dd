// Internal: the objects behind the C ABI's opaque handles and the helpers the translation units of
// the ABI layer (api.cpp, ring_api.cpp) share.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "device.hpp"
#include "sparse.hpp"
#include "triple.hpp"

// std::vector whose resize() leaves new doubles uninitialised (a blob of 1e8 doubles is written
// by several threads right after: value-initialising it first would touch every page twice)
template <class T>
struct default_init_alloc : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_alloc<U>; };
  template <class U, class... A> void construct(U *ptr, A &&...args) {
    if constexpr (sizeof...(A) == 0) ::new ((void *)ptr) U;
    else ::new ((void *)ptr) U(std::forward<A>(args)...);
  }
};
using BlobVec = std::vector<double, default_init_alloc<double>>;

struct cofactor_ctx {
  // Aggregates of one context share its stream and scratch buffers (partials, pair slabs, skip
  // list): every entry point that enqueues device work holds this lock for its whole sequence.
  std::recursive_mutex mu;
  int device = 0;
  hipStream_t stream = nullptr;
  int cus = 0;
  int gram_grid = 0;            // workgroups of the Gram kernel
  int cat_grid = 0;             // workgroups of the categorical kernel
  size_t lds_budget = 0;        // bytes of LDS one categorical workgroup may claim
  double *partials = nullptr;   // gram_grid * GRAM_ACC_LEN doubles
  unsigned *pair_slabs = nullptr;   // fused kernel: one u32 pair table per workgroup
  size_t pair_slab_bytes = 0;
  // optional HIP-event timing of the two streaming kernels (cofactor_ctx_profile_*)
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> gram_ev, cat_ev, fused_ev;
  const char *last_fused = "";  // name of the one-pass kernel launched last (cofactor_ctx_profile_kernel)
  bool allow_fused = true;      // COFACTOR_NO_FUSED=1 forces the two-kernel path
  int fused_pref = 0;           // COFACTOR_FUSED=1 / 2: only fused_kernel / only fused2_kernel (A/B runs);
                                // default: fused_kernel where it applies (faster at 10_10), else fused2_kernel
  bool allow_optimistic = true; // COFACTOR_NO_OPTIMISTIC=1: always run the dictionary pass first
  // experiment knobs, read once when the context is made (never on the update path)
  bool no_sub = false;          // COFACTOR_NO_SUB=1: no one-hot sub-launches on the code-cache route
  bool stage_split = false;     // COFACTOR_STAGE_SPLIT=1: one H2D copy per column instead of one per block
  uint64_t stage_rows_max = 1 << 18;   // COFACTOR_STAGE_ROWS: rows a state's staging buffer may grow to
  unsigned *skip = nullptr;     // optimistic fused pass: [count, tile ids...]
  size_t skip_bytes = 0;
  size_t lds_max = 160 * 1024;  // LDS one workgroup may claim on this device
  double *ring_red = nullptr;   // 256 doubles: reduced dense children of a vector of triples (sum_triple)
  unsigned char *predict_buf = nullptr;   // the model of a predict call on the device (keys, labels, weights), grown on demand
  size_t predict_bytes = 0;
  void *seg_scratch = nullptr;  // segmented GROUP BY (groupseg.hip): codes, offsets, regrouped records (grown on demand)
  size_t seg_scratch_bytes = 0;
  int groups_seg = 0;           // COFACTOR_GROUPS_SEG=1: segmented path whenever the shape allows, =2: never; default by size
  void *ring_scratch = nullptr; // multiply_triple: sub-list lengths / offsets / scan temporaries (grown on demand)
  size_t ring_scratch_bytes = 0;
  // multiply_triple: the plan a size query left in ring_scratch for the fill call that follows it
  bool mul_plan_valid = false;
  uintptr_t mul_plan_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t mul_plan_need[3] = {0, 0, 0};
  // multi-pass generic path: 16-bit key codes of the batch ([column][stride]) and the u32 cells of a
  // pair table too big for LDS
  unsigned short *code_cache = nullptr;
  size_t code_cache_bytes = 0;
  unsigned *pair_tmp = nullptr;
  size_t pair_tmp_bytes = 0;
  // binned pair tables (cat.hip): histogram / offset / cursor words and the regrouped code columns
  unsigned *bin_words = nullptr;
  cofactor::BinPlan *bin_plan = nullptr;   // device copy of the piece's plan
  unsigned short *bin_codes = nullptr;
  size_t bin_codes_bytes = 0;
  // finalize of states with millions of pair cells: the lists are written on the device into fin_dev
  // and land in the pinned host buffer fin_host (both grown on demand, kept); fin_owner / fin_epoch say
  // whose blob the pinned buffer currently holds
  double *fin_dev = nullptr, *fin_host = nullptr;
  size_t fin_dev_cap = 0, fin_host_cap = 0, fin_len = 0;
  const void *fin_owner = nullptr;
  bool allow_binned = true;     // COFACTOR_NO_BINNED=1: big pair tables by global atomics (round 2's path)
  cofactor::SparseScratch sparse_sc;   // sort / merge buffers of the sparse pair tables
  // Staging blocks (pinned + device, both buffers of a state) handed back by states that were
  // destroyed or outgrew them, reused by the next state of the same shape: the worker threads of
  // the next query do not pin 40 MB each again.
  struct StageBlock { int n, m; uint64_t cap; float *h_num, *d_num; int32_t *h_cat, *d_cat; };
  std::vector<StageBlock> stage_pool;
  size_t stage_pool_bytes = 0;
};

#define CTX_LOCK(ctxp) std::lock_guard<std::recursive_mutex> ctx_lock_((ctxp)->mu)

struct cofactor_agg {
  cofactor_ctx *ctx = nullptr;
  int n = 0, m = 0, kind = 0;
  cofactor::HostTriple host;    // everything merged in on the host (combine, lifted triples, import)
  double dev_rows = 0;          // rows (of unmasked updates) whose contributions sit in the device tables
  unsigned long long *d_kept = nullptr;   // device counter: rows kept by masked updates
  bool dev_dirty = false;       // the device tables hold something
  double *d_acc = nullptr;      // dense accumulator image (GRAM_ACC_LEN doubles)
  // categorical device state
  bool cat_ready = false;
  bool cat_check_pending = false;
  int32_t nkeys_host[COFACTOR_MAX_CAT] = {0};
  // finalize's two-call protocol: the blob of the size query is kept for the fill call
  BlobVec blob_cache;
  bool blob_cache_valid = false;
  bool blob_in_ctx = false;     // the valid blob sits in ctx->fin_host (device-encoded pair lists), not in blob_cache
  // pair tables kept as sorted lists (L.sparse_mask): one store per column pair, empty for dense pairs
  std::vector<cofactor::SparseStore> sparse;
  cofactor::CatLayout L{};
  cofactor::CatDevice D{};
  // host staging for update_host (pinned) and its device mirror, both double-buffered: while
  // buffer b is on its way to the device (copy + kernels, asynchronous), chunks land in b ^ 1
  uint64_t stage_cap = 0, stage_rows = 0;
  int stage_buf = 0;
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  bool stage_busy[2] = {false, false};
  float *h_num = nullptr;
  int32_t *h_cat = nullptr;
  float *d_num = nullptr;
  int32_t *d_cat = nullptr;
  // dense seam: the host-side dense addends on their way to the export kernel
  double *d_host_dense = nullptr;
  std::vector<double> host_dense_stage;
  // table seam: signature of the key lists the dictionaries were last aligned to (0 = the
  // dictionaries have changed since, or were never aligned)
  uint64_t dict_sig = 0;
};

namespace cofactor {
namespace detail {

cofactor_status fail(cofactor_status st, const std::string &msg);
cofactor_status hip_fail(hipError_t e, const char *what);
long env_long(const char *name, long dflt);
int next_pow2(int v);

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// categorical state management shared with the ring ops (api.cpp)
void cat_free(CatDevice &D);
bool cat_finish_layout(CatLayout &L, bool allow_sparse = false);
cofactor_status cat_alloc(const CatLayout &L, CatDevice &D, bool fresh_counters, hipStream_t st);
cofactor_status cat_regrow(cofactor_agg *a, const CatLayout &Lnew);
cofactor_status cat_prepare(cofactor_agg *a);
cofactor_status stage_flush(cofactor_agg *a);
// dictionary maintenance for one batch of keys: `insert` launches the kernel that puts the batch's
// keys into a->D (geometry a->L); unseen keys get codes, dictionaries and tables grow as needed
cofactor_status cat_dictionaries_with(cofactor_agg *a, const std::function<hipError_t()> &insert);
cofactor_status cat_dictionaries(cofactor_agg *a, const CatCols &cat, uint64_t rows);
cofactor_status emit_blob(const std::vector<double> &blob, double *out, uint64_t cap, uint64_t *needed);
cofactor_status emit_blob(const double *blob, size_t size, double *out, uint64_t cap, uint64_t *needed);

}  // namespace detail
}  // namespace cofactor

#define HIP_TRY(expr)                                                  \
  do {                                                                 \
    hipError_t e_ = (expr);                                            \
    if (e_ != hipSuccess) return cofactor::detail::hip_fail(e_, #expr); \
  } while (0)
