// cofactor_comm / cofactor_agg_allreduce: Triple::SumStateCombine (duckdb_extension/src/triple/sum/
// sum_state.cpp:10-114) ACROSS GPUs, below the C ABI — the "single RCCL all-reduce of the partial
// triple over xGMI" of the build's north star, with no torch and no MPI in the data path.
//
// One process per GPU.  RCCL is opened with dlopen the first time a communicator is made (the library
// itself links only libamdhip64, so a DuckDB process that never goes multi-GPU never loads it; in a
// python process that already mapped torch's librccl the loader hands back that copy).
//
// cofactor_agg_allreduce, on the context stream unless noted:
//   1. header exchange — ncclAllGather of a few words per rank: status of the local preparation,
//      dictionary signature, key count of every column, length of every sorted pair list.  All ranks
//      see every rank's status: a rank that cannot take part makes ALL ranks return an error instead
//      of leaving the others inside a collective for ever.
//   2. only when a signature differs (a rank met a new key since the last common alignment): padded
//      ncclAllGather of the key lists, every rank forms the same union and runs
//      cofactor_agg_align_keys (remap kernels); a second status exchange.
//   3. export kernels -> ONE ncclAllReduce(sum, double) of [N, lin, quad | cnt | s | p] -> import
//      kernels.  ~1.8 KB at 20_0, ~127 KB at 10_10 / 16 keys: latency-bound, no bucketing.
//   4. pair tables kept as sorted lists: padded all-gather of (key, count) lists, merged on every
//      rank (sort + reduce by key).
// Steady state (a MICE loop, a bench loop): once a state shape has been through 1-3 on this
// communicator, the next call is ONE collective — the same all-reduce with 9 more words: the
// dictionary signature in 16-bit pieces x and x^2 (all ranks hold the same signature exactly when
// world * sum(x^2) == (sum x)^2 for every piece) and a status word.  Only when that check fails (a
// rank met a new key, or could not prepare) does everybody fall back to 1-3; nothing has been
// imported by then.  Without key columns (20_0) there is nothing to check and no host
// synchronisation at all: export kernel -> ncclAllReduce -> import kernel on the stream.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "state.hpp"

using namespace cofactor;
using namespace cofactor::detail;

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names)
      if ((r.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!r.handle) { r.error = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
    auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("RCCL lacks ") + n; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return r;
}

cofactor_status nccl_fail(ncclResult_t e, const char *what) {
  Rccl &r = rccl();
  return fail(COFACTOR_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}

#define NCCL_TRY(expr)                                          \
  do {                                                          \
    ncclResult_t e_ = (expr);                                   \
    if (e_ != ncclSuccess) return nccl_fail(e_, #expr);         \
  } while (0)

}  // namespace

struct cofactor_comm {
  cofactor_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  // device scratch of the seam, grown on demand: [mine | everybody's] words of the small exchanges,
  // and the image that is all-reduced
  unsigned long long *d_words = nullptr;
  size_t words_cap = 0;
  double *d_image = nullptr;
  size_t image_cap = 0;
  // what the last complete exchange on this communicator agreed on (the same on every rank: they
  // all went through it): states of that shape take the one-collective path next time
  bool opt_valid = false;
  unsigned long long opt_shape = 0;
  size_t opt_tlen = 0;
};

namespace {

cofactor_status reserve_words(cofactor_comm *c, size_t words) {
  if (words <= c->words_cap) return COFACTOR_OK;
  HIP_TRY(hipStreamSynchronize(c->ctx->stream));
  (void)hipFree(c->d_words);
  c->d_words = nullptr; c->words_cap = 0;
  HIP_TRY(hipMalloc((void **)&c->d_words, words * 8));
  c->words_cap = words;
  return COFACTOR_OK;
}

// every rank's `n` words on every rank (host vectors); synchronises the stream
cofactor_status allgather_words(cofactor_comm *c, const std::vector<unsigned long long> &mine,
                                std::vector<unsigned long long> &all) {
  const size_t n = mine.size();
  all.assign(n * c->world, 0ull);
  if (n == 0) return COFACTOR_OK;
  cofactor_status s = reserve_words(c, n * (c->world + 1));
  if (s != COFACTOR_OK) return s;
  hipStream_t st = c->ctx->stream;
  HIP_TRY(hipMemcpyAsync(c->d_words, mine.data(), n * 8, hipMemcpyHostToDevice, st));
  NCCL_TRY(rccl().AllGather(c->d_words, c->d_words + n, n, ncclUint64, c->comm, st));
  HIP_TRY(hipMemcpyAsync(all.data(), c->d_words + n, n * c->world * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return COFACTOR_OK;
}

}  // namespace

extern "C" {

cofactor_status cofactor_comm_unique_id(void *id_out) {
  if (!id_out) return fail(COFACTOR_ERR_INVALID, "null argument");
  Rccl &r = rccl();
  if (!r.error.empty()) return fail(COFACTOR_ERR_UNSUPPORTED, r.error);
  static_assert(sizeof(ncclUniqueId) == COFACTOR_COMM_ID_BYTES, "the id is handed around as 128 bytes");
  ncclUniqueId id;
  NCCL_TRY(r.GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return COFACTOR_OK;
}

cofactor_status cofactor_comm_create(cofactor_ctx *ctx, const void *id, int rank, int world, cofactor_comm **out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return fail(COFACTOR_ERR_INVALID, "bad argument");
  Rccl &r = rccl();
  if (!r.error.empty()) return fail(COFACTOR_ERR_UNSUPPORTED, r.error);
  CTX_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  auto *c = new cofactor_comm();
  c->ctx = ctx; c->rank = rank; c->world = world;
  ncclResult_t e = r.CommInitRank(&c->comm, world, uid, rank);
  if (e != ncclSuccess) { delete c; return nccl_fail(e, "ncclCommInitRank"); }
  *out = c;
  return COFACTOR_OK;
}

void cofactor_comm_destroy(cofactor_comm *c) {
  if (!c) return;
  {
    CTX_LOCK(c->ctx);
    DeviceGuard guard(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
    (void)hipFree(c->d_words);
    (void)hipFree(c->d_image);
  }
  delete c;
}

cofactor_status cofactor_agg_allreduce(cofactor_agg *a, cofactor_comm *c) {
  if (!a || !c) return fail(COFACTOR_ERR_INVALID, "null argument");
  if (a->ctx != c->ctx) return fail(COFACTOR_ERR_INVALID, "allreduce: the state lives on another context than the communicator");
  CTX_LOCK(a->ctx);
  DeviceGuard guard(a->ctx->device);
  hipStream_t st = a->ctx->stream;
  const int m = a->m, world = c->world;
  const int np = m * (m + 1) / 2;

  const unsigned long long shape = ((unsigned long long)a->kind << 32) | ((unsigned long long)a->n << 16) | (unsigned long long)m;
  const size_t dlen = (size_t)cofactor_dense_len(a->n, (cofactor_kind)a->kind);
  cofactor_status s = COFACTOR_OK;
  auto reserve_image = [&](size_t doubles) -> cofactor_status {
    if (doubles <= c->image_cap) return COFACTOR_OK;
    HIP_TRY(hipStreamSynchronize(st));
    (void)hipFree(c->d_image);
    c->d_image = nullptr; c->image_cap = 0;
    HIP_TRY(hipMalloc((void **)&c->d_image, doubles * sizeof(double)));
    c->image_cap = doubles;
    return COFACTOR_OK;
  };
  // ---- 0. one collective when this shape has been agreed on before ----
  bool lists_possible = m > 0 && a->kind == COFACTOR_TRIPLE;
  if (c->opt_valid && c->opt_shape == shape) {
    constexpr int CHK = 9;                            // 4 x (x, x^2) of the signature's 16-bit pieces + status
    const size_t tl = c->opt_tlen, total = dlen + tl + (m > 0 ? CHK : 0);
    if ((s = reserve_image(total)) != COFACTOR_OK) return s;
    uint64_t sig0 = 0;
    cofactor_status prep = COFACTOR_OK;
    if (m > 0) prep = cofactor_agg_dict_signature(a, &sig0);
    if (prep == COFACTOR_OK) prep = cofactor_agg_export_dense_device(a, c->d_image);
    const bool mine_ok = prep == COFACTOR_OK && (m == 0 || (sig0 != 0 && (size_t)cofactor_agg_tables_len(a) == tl));
    if (m > 0) {
      if (mine_ok) prep = cofactor_agg_export_tables_device(a, c->d_image + dlen);
      else HIP_TRY(hipMemsetAsync(c->d_image + dlen, 0, tl * sizeof(double), st));
      double chk[CHK];
      for (int i = 0; i < 4; i++) {
        const double x = mine_ok ? (double)((sig0 >> (16 * i)) & 0xFFFFull) : (double)(1 + c->rank + 7 * i);   // (a rank that is out: pieces no other rank shares)
        chk[2 * i] = x; chk[2 * i + 1] = x * x;
      }
      chk[8] = (mine_ok && prep == COFACTOR_OK) ? 0.0 : 1.0;
      HIP_TRY(hipMemcpyAsync(c->d_image + dlen + tl, chk, sizeof(chk), hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));              // (chk lives on this stack frame)
    } else if (prep != COFACTOR_OK) {
      return prep;                                    // (nothing another rank could be waiting for has been skipped yet... see below)
    }
    NCCL_TRY(rccl().AllReduce(c->d_image, c->d_image, total, ncclDouble, ncclSum, c->comm, st));
    bool agreed = true;
    if (m > 0) {
      double chk[CHK];
      HIP_TRY(hipMemcpyAsync(chk, c->d_image + dlen + tl, sizeof(chk), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      agreed = chk[8] == 0.0;
      for (int i = 0; i < 4 && agreed; i++) agreed = (double)world * chk[2 * i + 1] == chk[2 * i] * chk[2 * i];
    }
    if (agreed) {
      if ((s = cofactor_agg_import_dense_device(a, c->d_image)) != COFACTOR_OK) return s;
      if (tl && (s = cofactor_agg_import_tables_device(a, c->d_image + dlen)) != COFACTOR_OK) return s;
      if (!lists_possible) return COFACTOR_OK;
      goto sorted_lists;                              // (their lengths are exchanged every time: they follow the data)
    }
    // not agreed: nothing was imported; every rank takes the full exchange below
  }
  {
  // ---- 1. header: [status, shape word, signature, key count per column] ----
  std::vector<std::vector<int32_t>> own;
  cofactor_status local = COFACTOR_OK;
  std::string local_msg;
  uint64_t sig = 0;
  if (m > 0) {
    local = cofactor_agg_dict_signature(a, &sig);
    if (local == COFACTOR_OK && sig == 0) {
      uint64_t need = 0;
      std::vector<uint64_t> offs(m + 1, 0);
      local = cofactor_agg_keys(a, nullptr, 0, &need, nullptr);
      std::vector<int32_t> flat(std::max<uint64_t>(1, need));
      if (local == COFACTOR_OK) local = cofactor_agg_keys(a, flat.data(), flat.size(), &need, offs.data());
      own.assign(m, {});
      if (local == COFACTOR_OK)
        for (int col = 0; col < m; col++) own[col].assign(flat.begin() + offs[col], flat.begin() + offs[col + 1]);
    }
    if (local != COFACTOR_OK) local_msg = cofactor_last_error();
  }
  std::vector<unsigned long long> head(3 + m, 0ull), heads;
  head[0] = (unsigned long long)local;
  head[1] = shape;
  head[2] = sig;
  for (int col = 0; col < m; col++) head[3 + col] = sig == 0 && !own.empty() ? own[col].size() : 0;
  s = allgather_words(c, head, heads);
  if (s != COFACTOR_OK) return s;
  const size_t hw = head.size();
  for (int r = 0; r < world; r++) {
    if (heads[r * hw + 0] != 0)
      return fail(COFACTOR_ERR_INVALID, r == c->rank ? "allreduce: this rank could not prepare its state: " + local_msg
                                                    : "allreduce: rank " + std::to_string(r) + " could not prepare its state");
    if (heads[r * hw + 1] != shape) return fail(COFACTOR_ERR_INVALID, "allreduce: the ranks' states differ in kind / n / m");
  }
  // ---- 2. dictionaries: gather the key lists when some rank left the common alignment ----
  if (m > 0) {
    bool same = sig != 0;
    for (int r = 0; r < world && same; r++) same = heads[r * hw + 2] == sig;
    if (!same) {
      if (own.empty()) {                              // this rank was aligned: it still has to list its keys
        uint64_t need = 0;
        std::vector<uint64_t> offs(m + 1, 0);
        local = cofactor_agg_keys(a, nullptr, 0, &need, nullptr);
        std::vector<int32_t> flat(std::max<uint64_t>(1, need));
        if (local == COFACTOR_OK) local = cofactor_agg_keys(a, flat.data(), flat.size(), &need, offs.data());
        own.assign(m, {});
        if (local == COFACTOR_OK)
          for (int col = 0; col < m; col++) own[col].assign(flat.begin() + offs[col], flat.begin() + offs[col + 1]);
      }
      // sizes of everybody's lists, then the lists themselves padded to the longest
      std::vector<unsigned long long> sz(1 + m, 0ull), szs;
      sz[0] = (unsigned long long)local;
      for (int col = 0; col < m && local == COFACTOR_OK; col++) sz[1 + col] = own[col].size();
      if ((s = allgather_words(c, sz, szs)) != COFACTOR_OK) return s;
      size_t longest = 1;
      for (int r = 0; r < world; r++) {
        if (szs[r * (1 + m)] != 0) return fail(COFACTOR_ERR_INVALID, "allreduce: rank " + std::to_string(r) + " could not list its keys");
        size_t tot = 0;
        for (int col = 0; col < m; col++) tot += szs[r * (1 + m) + 1 + col];
        longest = std::max(longest, tot);
      }
      std::vector<unsigned long long> mine(longest, 0ull), lists;
      {
        size_t pos = 0;
        for (int col = 0; col < m; col++)
          for (int32_t k : own[col]) mine[pos++] = (unsigned long long)(uint32_t)k;
      }
      if ((s = allgather_words(c, mine, lists)) != COFACTOR_OK) return s;
      std::vector<int32_t> all;
      std::vector<uint64_t> offs(m + 1, 0);
      for (int col = 0; col < m; col++) {
        for (int r = 0; r < world; r++) {
          size_t pos = 0;
          for (int c2 = 0; c2 < col; c2++) pos += szs[r * (1 + m) + 1 + c2];
          const size_t cnt = szs[r * (1 + m) + 1 + col];
          for (size_t i = 0; i < cnt; i++) all.push_back((int32_t)(uint32_t)lists[r * longest + pos + i]);
        }
        offs[col + 1] = all.size();
      }
      local = cofactor_agg_align_keys(a, all.data(), offs.data());
      // agree that every rank is aligned before anybody enters the big all-reduce
      std::vector<unsigned long long> ok(1, (unsigned long long)local), oks;
      if (local != COFACTOR_OK) local_msg = cofactor_last_error();
      if ((s = allgather_words(c, ok, oks)) != COFACTOR_OK) return s;
      for (int r = 0; r < world; r++)
        if (oks[r] != 0)
          return fail(COFACTOR_ERR_INVALID, r == c->rank ? "allreduce: aligning the dictionaries failed on this rank: " + local_msg
                                                        : "allreduce: aligning the dictionaries failed on rank " + std::to_string(r));
    }
  }
  // ---- 3. ONE all-reduce of [N, lin, quad | cnt | s | p] ----
  const size_t tlen = (size_t)cofactor_agg_tables_len(a);
  {
    // (table lengths follow from the aligned key lists: equal on all ranks by construction; checked all the same)
    std::vector<unsigned long long> ln(1, (unsigned long long)tlen), lns;
    if ((s = allgather_words(c, ln, lns)) != COFACTOR_OK) return s;
    for (int r = 0; r < world; r++)
      if (lns[r] != tlen) return fail(COFACTOR_ERR_INTERNAL, "allreduce: the ranks' aligned tables differ in size");
  }
  if ((s = reserve_image(dlen + tlen)) != COFACTOR_OK) return s;
  if ((s = cofactor_agg_export_dense_device(a, c->d_image)) != COFACTOR_OK) return s;
  if (tlen && (s = cofactor_agg_export_tables_device(a, c->d_image + dlen)) != COFACTOR_OK) return s;
  NCCL_TRY(rccl().AllReduce(c->d_image, c->d_image, dlen + tlen, ncclDouble, ncclSum, c->comm, st));
  if ((s = cofactor_agg_import_dense_device(a, c->d_image)) != COFACTOR_OK) return s;
  if (tlen && (s = cofactor_agg_import_tables_device(a, c->d_image + dlen)) != COFACTOR_OK) return s;
  c->opt_valid = true; c->opt_shape = shape; c->opt_tlen = tlen;
  }
sorted_lists:
  // ---- 4. sorted pair lists: gather everybody's, merge ----
  if (lists_possible) {
    std::vector<uint64_t> lens(np, 0);
    if ((s = cofactor_agg_sparse_lens(a, lens.data(), lens.size())) != COFACTOR_OK) return s;
    std::vector<unsigned long long> mine(lens.begin(), lens.end()), alls;
    if ((s = allgather_words(c, mine, alls)) != COFACTOR_OK) return s;
    for (int q = 0; q < np; q++) {
      size_t longest = 0, total = 0;
      for (int r = 0; r < world; r++) { longest = std::max<size_t>(longest, alls[(size_t)r * np + q]); total += alls[(size_t)r * np + q]; }
      if (total == 0) continue;
      // [mine: keys | counts, padded to `longest` each] -> all-gather -> compact -> assign
      if ((s = reserve_words(c, 2 * longest * (world + 1) + 2 * total)) != COFACTOR_OK) return s;
      unsigned long long *send = c->d_words, *recv = c->d_words + 2 * longest, *flat = recv + 2 * longest * world;
      HIP_TRY(hipMemsetAsync(send, 0, 2 * longest * 8, st));
      if ((s = cofactor_agg_sparse_export_device(a, q, (uint64_t *)send, (uint64_t *)(send + longest))) != COFACTOR_OK) return s;
      NCCL_TRY(rccl().AllGather(send, recv, 2 * longest, ncclUint64, c->comm, st));
      size_t pos = 0;
      for (int r = 0; r < world; r++) {
        const size_t len = alls[(size_t)r * np + q];
        if (!len) continue;
        HIP_TRY(hipMemcpyAsync(flat + pos, recv + (size_t)r * 2 * longest, len * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(flat + total + pos, recv + (size_t)r * 2 * longest + longest, len * 8, hipMemcpyDeviceToDevice, st));
        pos += len;
      }
      if ((s = cofactor_agg_sparse_assign_device(a, q, (const uint64_t *)flat, (const uint64_t *)(flat + total), total)) != COFACTOR_OK)
        return s;
    }
  }
  return COFACTOR_OK;
}

}  // extern "C"
