// Sparse pair tables (see sparse.hpp): sort + run-length encode the key pairs of a batch, then
// merge the runs into the sorted store (concatenate, sort by key, reduce by key).  The sorts are
// hipCUB radix sorts; the kernels here only form the 64-bit keys and compact table cells.
#include "sparse.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>

namespace cofactor {
namespace {

__device__ __forceinline__ unsigned long long pack_dev(int32_t k1, int32_t k2) {
  return ((unsigned long long)((uint32_t)k1 ^ 0x80000000u) << 32) | (unsigned long long)((uint32_t)k2 ^ 0x80000000u);
}

// out[i] = packed (col1[i], col2[i]); with a mask the kept rows are compacted (any order: they are
// sorted next), one atomic per wave
__global__ __launch_bounds__(256) void pair_keys_kernel(const int32_t *__restrict__ c1, const int32_t *__restrict__ c2,
                                                        const uint8_t *__restrict__ mask, uint64_t rows,
                                                        unsigned long long *__restrict__ out,
                                                        unsigned long long *__restrict__ counter) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  // every lane of a wave runs the same number of iterations (the ballot below needs them all)
  const uint64_t iters = (rows + step - 1) / step;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t it = 0; it < iters; it++, i += step) {
    const bool in = i < rows;
    if (!mask) {
      if (in) out[i] = pack_dev(c1[i], c2[i]);
      continue;
    }
    const bool keep = in && mask[i] != 0;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
    if (b == 0ull) continue;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned long long)__builtin_popcountll(b));
    base = __shfl(base, 0, 64);
    if (keep) out[base + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = pack_dev(c1[i], c2[i]);
  }
}

__global__ __launch_bounds__(256) void widen_kernel(const unsigned *__restrict__ in, unsigned long long *__restrict__ out,
                                                    uint64_t count) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = in[i];
}

__global__ __launch_bounds__(256) void dense_cells_kernel(const unsigned long long *__restrict__ table, int kc1, int kc2,
                                                          const int32_t *__restrict__ key_of1,
                                                          const int32_t *__restrict__ key_of2,
                                                          unsigned long long *__restrict__ keys,
                                                          unsigned long long *__restrict__ vals,
                                                          unsigned long long *__restrict__ counter) {
  const uint64_t cells = (uint64_t)kc1 * (uint64_t)kc2;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t iters = (cells + step - 1) / step;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t it = 0; it < iters; it++, i += step) {
    const unsigned long long v = i < cells ? table[i] : 0ull;
    const bool keep = v != 0ull;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
    if (b == 0ull) continue;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned long long)__builtin_popcountll(b));
    base = __shfl(base, 0, 64);
    if (keep) {
      const uint64_t at = base + __builtin_popcountll(b & ((1ull << lane) - 1ull));
      keys[at] = pack_dev(key_of1[i / kc2], key_of2[i % kc2]);
      vals[at] = v;
    }
  }
}

__global__ __launch_bounds__(256) void key_of_code_kernel(const unsigned long long *__restrict__ slots,
                                                          const int32_t *__restrict__ codes, int cap, int kc,
                                                          int32_t *__restrict__ key_of) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += gridDim.x * blockDim.x) {
    const unsigned long long v = slots[s];
    const int32_t cd = codes[s];
    if (v != 0ull && cd >= 0 && cd < kc) key_of[cd] = (int32_t)(unsigned)(v & 0xffffffffull);
  }
}

hipError_t reserve(SparseScratch &sc, int which, size_t bytes, hipStream_t stream) {
  if (bytes <= sc.bytes[which]) return hipSuccess;
  hipError_t e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return e;
  (void)hipFree(sc.buf[which]);
  sc.buf[which] = nullptr;
  sc.bytes[which] = 0;
  bytes += bytes / 8;
  if ((e = hipMalloc(&sc.buf[which], bytes)) != hipSuccess) return e;
  sc.bytes[which] = bytes;
  return hipSuccess;
}

int grid_for(uint64_t items) { return (int)std::min<uint64_t>(std::max<uint64_t>(1, (items + 255) / 256), 4096); }

hipError_t ensure_counter(SparseScratch &sc) {
  if (sc.counter) return hipSuccess;
  return hipMalloc((void **)&sc.counter, 2 * sizeof(unsigned long long));
}

// st += (keys[i], vals[i]), i < count; keys in any order, duplicates allowed.  keys / vals are
// scratch buffers 0 / 1 and are consumed.  sorted_unique: they are ascending and unique already.
hipError_t merge_into(SparseScratch &sc, SparseStore &st, unsigned long long *keys, unsigned long long *vals, size_t count,
                      bool sorted_unique, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  hipError_t e;
  const size_t total = st.len + count;
  if (total > 0x7fffffffull) return hipErrorInvalidValue;    // (item counts of the primitives are ints)
  unsigned long long *nk = nullptr, *nv = nullptr;
  if ((e = hipMalloc((void **)&nk, total * 8)) != hipSuccess) return e;
  if ((e = hipMalloc((void **)&nv, total * 8)) != hipSuccess) { (void)hipFree(nk); return e; }
  size_t new_len = 0;
  auto fail = [&](hipError_t err) { (void)hipFree(nk); (void)hipFree(nv); return err; };
  if (st.len == 0 && sorted_unique) {
    if ((e = hipMemcpyAsync(nk, keys, count * 8, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(nv, vals, count * 8, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return fail(e);
    new_len = count;
  } else {
    // [store | batch] in scratch 2 / 3, sorted into 4 / 5, reduced into the new store
    if ((e = reserve(sc, 2, total * 8, stream)) != hipSuccess) return fail(e);
    if ((e = reserve(sc, 3, total * 8, stream)) != hipSuccess) return fail(e);
    if ((e = reserve(sc, 4, total * 8, stream)) != hipSuccess) return fail(e);
    if ((e = reserve(sc, 5, total * 8, stream)) != hipSuccess) return fail(e);
    auto *ck = (unsigned long long *)sc.buf[2], *cv = (unsigned long long *)sc.buf[3];
    auto *sk = (unsigned long long *)sc.buf[4], *sv = (unsigned long long *)sc.buf[5];
    if (st.len) {
      if ((e = hipMemcpyAsync(ck, st.keys, st.len * 8, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return fail(e);
      if ((e = hipMemcpyAsync(cv, st.cnt, st.len * 8, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return fail(e);
    }
    if ((e = hipMemcpyAsync(ck + st.len, keys, count * 8, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(cv + st.len, vals, count * 8, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return fail(e);
    size_t t1 = 0, t2 = 0;
    if ((e = hipcub::DeviceRadixSort::SortPairs(nullptr, t1, ck, sk, cv, sv, (int)total, 0, 64, stream)) != hipSuccess) return fail(e);
    if ((e = hipcub::DeviceReduce::ReduceByKey(nullptr, t2, sk, nk, sv, nv, sc.counter, hipcub::Sum(), (int)total, stream)) != hipSuccess)
      return fail(e);
    // the sort's temporaries may overlay the batch buffers (0 / 1): they were copied above
    const size_t tmax = std::max(t1, t2);
    if ((e = reserve(sc, 0, std::max(tmax, sc.bytes[0]), stream)) != hipSuccess) return fail(e);
    if ((e = hipcub::DeviceRadixSort::SortPairs(sc.buf[0], t1, ck, sk, cv, sv, (int)total, 0, 64, stream)) != hipSuccess) return fail(e);
    if ((e = hipcub::DeviceReduce::ReduceByKey(sc.buf[0], t2, sk, nk, sv, nv, sc.counter, hipcub::Sum(), (int)total, stream)) != hipSuccess)
      return fail(e);
    unsigned long long runs = 0;
    if ((e = hipMemcpyAsync(&runs, sc.counter, 8, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e);
    new_len = (size_t)runs;
  }
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e);
  (void)hipFree(st.keys);
  (void)hipFree(st.cnt);
  st.keys = nk; st.cnt = nv; st.len = new_len; st.cap = total;
  return hipSuccess;
}

}  // namespace

void sparse_store_free(SparseStore &st) {
  (void)hipFree(st.keys);
  (void)hipFree(st.cnt);
  st = SparseStore{};
}

void sparse_scratch_free(SparseScratch &sc) {
  for (auto &b : sc.buf) { (void)hipFree(b); b = nullptr; }
  for (auto &b : sc.bytes) b = 0;
  (void)hipFree(sc.counter);
  sc.counter = nullptr;
}

hipError_t sparse_add_rows(SparseScratch &sc, SparseStore &st, const int32_t *col1, const int32_t *col2,
                           const uint8_t *mask, uint64_t rows, hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  if (rows > 0x7fffffffull) return hipErrorInvalidValue;
  hipError_t e;
  if ((e = ensure_counter(sc)) != hipSuccess) return e;
  if ((e = reserve(sc, 2, rows * 8, stream)) != hipSuccess) return e;
  if ((e = reserve(sc, 3, rows * 8, stream)) != hipSuccess) return e;
  auto *raw = (unsigned long long *)sc.buf[2], *sorted = (unsigned long long *)sc.buf[3];
  if ((e = hipMemsetAsync(sc.counter, 0, 16, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(pair_keys_kernel, dim3(grid_for(rows)), dim3(256), 0, stream, col1, col2, mask, rows, raw, sc.counter);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  uint64_t kept = rows;
  if (mask) {
    unsigned long long c = 0;
    if ((e = hipMemcpyAsync(&c, sc.counter, 8, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    kept = c;
    if (kept == 0) return hipSuccess;
  }
  // sort, then runs: unique keys -> scratch 0, lengths (u32) -> scratch 4, widened -> scratch 1
  size_t t1 = 0, t2 = 0;
  if ((e = reserve(sc, 0, kept * 8, stream)) != hipSuccess) return e;
  if ((e = reserve(sc, 1, kept * 8, stream)) != hipSuccess) return e;
  if ((e = reserve(sc, 4, kept * 4, stream)) != hipSuccess) return e;
  auto *ukeys = (unsigned long long *)sc.buf[0];
  auto *uvals = (unsigned long long *)sc.buf[1];
  auto *lens = (unsigned *)sc.buf[4];
  if ((e = hipcub::DeviceRadixSort::SortKeys(nullptr, t1, raw, sorted, (int)kept, 0, 64, stream)) != hipSuccess) return e;
  if ((e = hipcub::DeviceRunLengthEncode::Encode(nullptr, t2, sorted, ukeys, lens, sc.counter + 1, (int)kept, stream)) != hipSuccess)
    return e;
  if ((e = reserve(sc, 5, std::max(t1, t2), stream)) != hipSuccess) return e;
  if ((e = hipcub::DeviceRadixSort::SortKeys(sc.buf[5], t1, raw, sorted, (int)kept, 0, 64, stream)) != hipSuccess) return e;
  if ((e = hipcub::DeviceRunLengthEncode::Encode(sc.buf[5], t2, sorted, ukeys, lens, sc.counter + 1, (int)kept, stream)) != hipSuccess)
    return e;
  unsigned long long runs = 0;
  if ((e = hipMemcpyAsync(&runs, sc.counter + 1, 8, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  // (the run counter is written as the iterator's value type: an unsigned long long here)
  hipLaunchKernelGGL(widen_kernel, dim3(grid_for(runs)), dim3(256), 0, stream, lens, uvals, (uint64_t)runs);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return merge_into(sc, st, ukeys, uvals, (size_t)runs, true, stream);
}

hipError_t sparse_add_dense(SparseScratch &sc, SparseStore &st, const unsigned long long *table, int kc1, int kc2,
                            const int32_t *key_of1, const int32_t *key_of2, hipStream_t stream) {
  const uint64_t cells = (uint64_t)kc1 * (uint64_t)kc2;
  if (cells == 0) return hipSuccess;
  if (cells > 0x7fffffffull) return hipErrorInvalidValue;
  hipError_t e;
  if ((e = ensure_counter(sc)) != hipSuccess) return e;
  if ((e = reserve(sc, 0, cells * 8, stream)) != hipSuccess) return e;
  if ((e = reserve(sc, 1, cells * 8, stream)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(sc.counter, 0, 16, stream)) != hipSuccess) return e;
  auto *keys = (unsigned long long *)sc.buf[0], *vals = (unsigned long long *)sc.buf[1];
  hipLaunchKernelGGL(dense_cells_kernel, dim3(grid_for(cells)), dim3(256), 0, stream, table, kc1, kc2, key_of1, key_of2, keys,
                     vals, sc.counter);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  unsigned long long c = 0;
  if ((e = hipMemcpyAsync(&c, sc.counter, 8, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  return merge_into(sc, st, keys, vals, (size_t)c, false, stream);
}

hipError_t sparse_merge_lists(SparseScratch &sc, SparseStore &st, const unsigned long long *keys,
                              const unsigned long long *cnt, size_t len, hipStream_t stream) {
  if (len == 0) return hipSuccess;
  hipError_t e;
  if ((e = ensure_counter(sc)) != hipSuccess) return e;
  if ((e = reserve(sc, 0, len * 8, stream)) != hipSuccess) return e;
  if ((e = reserve(sc, 1, len * 8, stream)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(sc.buf[0], keys, len * 8, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(sc.buf[1], cnt, len * 8, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return e;
  return merge_into(sc, st, (unsigned long long *)sc.buf[0], (unsigned long long *)sc.buf[1], len, /*sorted_unique=*/false, stream);
}

hipError_t launch_key_of_code(const unsigned long long *slots, const int32_t *codes, int cap, int kc, int32_t *key_of,
                              hipStream_t stream) {
  hipLaunchKernelGGL(key_of_code_kernel, dim3(grid_for((uint64_t)cap)), dim3(256), 0, stream, slots, codes, cap, kc, key_of);
  return hipGetLastError();
}

}  // namespace cofactor
