// Sparse pair tables: the (key1, key2) -> count table of two high-cardinality key columns, kept as
// a sorted list in device memory instead of a dense code-indexed table.
//
// The reference holds every pair table in a std::map<std::pair<int,int>, float>
// (duckdb_extension/src/triple/sum/sum_no_lift.cpp:195-214): memory follows the number of pairs
// that occur, not the product of the cardinalities.  The dense tables of this library follow the
// product; a pair of columns whose product passes the threshold in api.cpp (cat_finish_layout)
// goes here instead.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace cofactor {

// 64-bit sort key of a key pair: ascending order of the word = ascending (key1, key2) as signed ints
inline unsigned long long sparse_pack(int32_t k1, int32_t k2) {
  return ((unsigned long long)((uint32_t)k1 ^ 0x80000000u) << 32) | (unsigned long long)((uint32_t)k2 ^ 0x80000000u);
}
inline void sparse_unpack(unsigned long long w, int32_t &k1, int32_t &k2) {
  k1 = (int32_t)((uint32_t)(w >> 32) ^ 0x80000000u);
  k2 = (int32_t)((uint32_t)(w & 0xffffffffull) ^ 0x80000000u);
}

struct SparseStore {              // sorted, unique
  unsigned long long *keys = nullptr;
  unsigned long long *cnt = nullptr;
  size_t len = 0, cap = 0;
};

struct SparseScratch {            // per context, grown on demand, reused by every call
  void *buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t bytes[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long *counter = nullptr;   // one device word
};

void sparse_store_free(SparseStore &st);
void sparse_scratch_free(SparseScratch &sc);

// st += the key pairs of rows [0, rows) of two key columns (rows whose mask byte is 0 left out).
// Synchronises the stream.  rows < 2^31.
hipError_t sparse_add_rows(SparseScratch &sc, SparseStore &st, const int32_t *col1, const int32_t *col2,
                           const uint8_t *mask, uint64_t rows, hipStream_t stream);

// st += the non-zero cells of a dense code-indexed table [kc1][kc2]; key_of1 / key_of2 (device):
// code -> key of the two columns.
hipError_t sparse_add_dense(SparseScratch &sc, SparseStore &st, const unsigned long long *table, int kc1, int kc2,
                            const int32_t *key_of1, const int32_t *key_of2, hipStream_t stream);

// st += (keys[i], cnt[i]), i < len (device arrays of this device, any order, duplicates allowed; the
// arrays are only read): another state's store (combine), or the gathered stores of all ranks.
hipError_t sparse_merge_lists(SparseScratch &sc, SparseStore &st, const unsigned long long *keys,
                              const unsigned long long *cnt, size_t len, hipStream_t stream);

// key_of[code] = key for one column's dictionary (slots / codes of that column, cap slots)
hipError_t launch_key_of_code(const unsigned long long *slots, const int32_t *codes, int cap, int kc, int32_t *key_of,
                              hipStream_t stream);

}  // namespace cofactor
