// Internal device-side declarations shared by the HIP translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/cofactor_hip.h"

namespace cofactor {

// ---- dense Gram kernel geometry ---------------------------------------------------------------
// One v_mfma_f32_4x4x1_16b_f32 multiplies 16 independent 4x1 by 1x4 blocks.  The n <= 20 numeric
// columns are cut into NB = ceil(n/4) <= 5 column blocks, so the upper triangle of the n x n Gram
// matrix is NB(NB+1)/2 <= 15 block pairs: ONE MFMA per input row.  Lane l = 4*b + t serves block
// pair b = (bi, bj): it feeds x[4*bi + t] as A and x[4*bj + t] as B and receives, in accumulator
// register i, the sum over rows of x[4*bi + i] * x[4*bj + t].
constexpr int GRAM_TILE_ROWS = 256;               // rows per LDS tile
constexpr int GRAM_COL_STRIDE = GRAM_TILE_ROWS + 4;  // floats; +4 keeps 16-B alignment and walks
                                                     // the 64 ds_read_b128 banks 4 per column
constexpr int GRAM_THREADS = 256;
constexpr int GRAM_SLOTS = 5;                     // per lane: 4 accumulator rows + 1 column sum
constexpr int GRAM_ACC_LEN = 64 * GRAM_SLOTS;     // doubles in one dense accumulator image

struct NumCols { const float *p[COFACTOR_MAX_NUM]; };
struct CatCols { const int32_t *p[COFACTOR_MAX_CAT]; };

__host__ __device__ inline int gram_pair_index(int bi, int bj, int nb) {  // bi <= bj
  return bi * nb - bi * (bi - 1) / 2 + (bj - bi);
}
// where Q[j][k] (j <= k) and lin[c] live in an accumulator image ([slot][lane])
__host__ __device__ inline int gram_quad_pos(int j, int k, int n) {
  const int nb = (n + 3) / 4;
  return (j & 3) * 64 + 4 * gram_pair_index(j >> 2, k >> 2, nb) + (k & 3);
}
__host__ __device__ inline int gram_lin_pos(int c, int n) {
  const int nb = (n + 3) / 4;
  return 4 * 64 + 4 * gram_pair_index(c >> 2, c >> 2, nb) + (c & 3);
}

// Launches the Gram pass over `rows` rows of n columns and adds the result into acc[GRAM_ACC_LEN]
// (device doubles).  partials must hold grid * GRAM_ACC_LEN doubles.
// ev0 / ev1 (optional) are recorded on `stream` right before / after the Gram kernel itself.
// mask (optional, one byte per row): rows whose byte is 0 contribute nothing.
hipError_t launch_gram(const NumCols &cols, int n, uint64_t rows, int grid, double *partials,
                       double *acc, hipStream_t stream, hipEvent_t ev0 = nullptr,
                       hipEvent_t ev1 = nullptr, const uint8_t *mask = nullptr);
hipError_t launch_count_mask(const uint8_t *mask, uint64_t rows, unsigned long long *counter, hipStream_t stream);

// adds the per-workgroup images partials[slot][wg] into acc[slot] in a fixed order
hipError_t launch_gram_fold(const double *partials, int nwg, double *acc, hipStream_t stream);

// Dense seam of the multi-GPU path: [N, lin, quad] <-> accumulator image, on the device.
// n_base: rows counted on the host side (unmasked updates); extra (optional): dense part the state
// holds on the host, added element-wise.
hipError_t launch_dense_export(const double *acc, const unsigned long long *kept, double n_base,
                               const double *extra, int n, int kind, double *out, hipStream_t stream);
hipError_t launch_dense_import(const double *in, int n, int kind, double *acc, unsigned long long *kept,
                               hipStream_t stream);

// acc[GRAM_ACC_LEN] += src_acc[..], *kept += *src_kept (all on this device)
hipError_t launch_acc_add(const double *src_acc, const unsigned long long *src_kept, double *acc,
                          unsigned long long *kept, hipStream_t stream);

// calibration kernels (cofactor_ctx_calibrate): float4 copy / read-only stream over `bytes`
hipError_t launch_calibration(const void *src, void *dst, uint64_t bytes, int grid, bool copy, hipStream_t stream);
uint64_t calibration_bytes(uint64_t bytes, int grid);   // bytes one launch actually reads

// ---- categorical tables -----------------------------------------------------------------------
constexpr int MAX_PAIRS = COFACTOR_MAX_CAT * (COFACTOR_MAX_CAT + 1) / 2;

// Everything a kernel needs to know about where a column's dictionary and tables live.
struct CatLayout {
  int n, m, kind;
  int ht_cap[COFACTOR_MAX_CAT];   // dictionary slots per column (power of two)
  int ht_off[COFACTOR_MAX_CAT];   // first slot of the column in ht_slot / ht_code
  int kc[COFACTOR_MAX_CAT];       // code capacity of the column (>= its number of keys)
  int cnt_off[COFACTOR_MAX_CAT];  // counts:  cnt[cnt_off[c] + code]
  int s_off[COFACTOR_MAX_CAT];    // sums:    s[s_off[c] + code * n + k]
  int p_off[MAX_PAIRS];           // pairs:   p[p_off[q] + code1 * kc[c2] + code2]
  int n_slots, n_cnt, n_s, n_p;
  // pair tables kept as sorted (key1, key2) -> count lists outside this layout (sparse.hpp): no
  // cells in p (their p_off equals the next pair's)
  unsigned sparse_mask[(MAX_PAIRS + 31) / 32];
};
__host__ __device__ inline bool pair_is_sparse(const CatLayout &L, int q) { return (L.sparse_mask[q >> 5] >> (q & 31)) & 1u; }
inline bool any_sparse_pair(const CatLayout &L) {
  for (unsigned w : L.sparse_mask) if (w) return true;
  return false;
}

struct CatDevice {
  unsigned long long *ht_slot;    // (1 << 32 | (uint32)key), 0 = empty
  int32_t *ht_code;               // code of the key in that slot, -1 = not assigned yet
  int32_t *nkeys;                 // per column number of keys (device counters)
  int32_t *flags;                 // [0] = a dictionary ran full during insert
  unsigned long long *cnt;
  double *s;
  unsigned long long *p;
};

// What one cat_accumulate launch updates.
struct CatPass {
  unsigned pair_mask[(MAX_PAIRS + 31) / 32];  // pair tables of this pass
  unsigned col_mask;                          // key columns the pass has to read
  int do_cnt, do_s;                           // counts / per-key sums in this pass
  int p_base, p_cells;                        // LDS pair tables cover cells [p_base, p_base + p_cells)
  int dict_lds;                               // dictionaries are copied to LDS
};
// Keys 0..255 of a column (the usual encoding of a categorical column) skip the hash probe: next to
// the LDS copy of the dictionaries sits a u16 table per column, key -> code, with one more entry
// that every other key reads (0xFFFF = not in the table).
constexpr int CAT_DIRECT_KEYS = 256;
constexpr int CAT_DIRECT_STRIDE = 260;            // u16 entries per column
inline size_t cat_direct_lds_bytes(int m) { return (size_t)m * CAT_DIRECT_STRIDE * 2 + 32; }
size_t cat_pass_lds_bytes(const CatLayout &L, const CatPass &P, bool lds_tables);

hipError_t launch_cat_insert(const CatCols &cols, uint64_t rows, const CatLayout &L,
                             const CatDevice &D, hipStream_t stream);
hipError_t launch_cat_assign_codes(const CatLayout &L, const CatDevice &D, hipStream_t stream);
hipError_t launch_cat_accumulate(const NumCols &num, const CatCols &cat, uint64_t rows,
                                 const CatLayout &L, const CatDevice &D, const CatPass &P,
                                 bool lds_tables, int grid, hipStream_t stream,
                                 hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                                 const uint8_t *mask = nullptr);
// re-inserts every (key, code) of the old dictionary into the new one (dictionary growth)
hipError_t launch_cat_rehash(const CatLayout &Lold, const CatDevice &Dold, const CatLayout &Lnew,
                             const CatDevice &Dnew, hipStream_t stream);
// copies the count / sum / pair tables from the old strides to the new ones (code growth)
hipError_t launch_cat_relayout(const CatLayout &Lold, const CatDevice &Dold, const CatLayout &Lnew,
                               const CatDevice &Dnew, hipStream_t stream);

// Multi-pass generic path: keys -> 16-bit codes once ([column][stride], stride = rows rounded up to 4),
// then count / sum passes over column subsets and pair passes that read the codes (cat.hip).
hipError_t launch_cat_codes(const CatCols &cat, uint64_t rows, uint64_t stride, const CatLayout &L, const CatDevice &D,
                            const uint8_t *mask, unsigned short *codes, hipStream_t stream, bool optimistic = false);
size_t cat_sums_lds_bytes(const CatLayout &L, unsigned col_mask, bool do_s);
hipError_t launch_cat_sums(const NumCols &num, const unsigned short *codes, uint64_t rows, uint64_t stride,
                           const CatLayout &L, const CatDevice &D, unsigned col_mask, int grid, hipStream_t stream);
// several column subsets (each fits LDS on its own) in ONE launch: the rows are read from HBM once
// the same for key columns of 17 .. 64 codes, on the matrix cores (catsums.hip)
bool cat_sums_mfma_applicable(const CatLayout &L, unsigned col_mask, uint64_t rows);
hipError_t launch_cat_sums_mfma(const NumCols &num, const unsigned short *codes, uint64_t rows, uint64_t stride,
                                const CatLayout &L, const CatDevice &D, unsigned col_mask, int wgs, hipStream_t stream);
hipError_t launch_cat_sums_subsets(const NumCols &num, const unsigned short *codes, uint64_t rows, uint64_t stride,
                                   const CatLayout &L, const CatDevice &D, const unsigned *masks, int nsub, int cus,
                                   hipStream_t stream);
// gtab == nullptr: LDS tables for the pairs of P (cells [p_base, p_base + p_cells)); else ONE pair, u32 cells in gtab
hipError_t launch_cat_pairs(const unsigned short *codes, uint64_t rows, uint64_t stride, const CatLayout &L, const CatDevice &D,
                            const CatPass &P, unsigned *gtab, int grid, hipStream_t stream);
hipError_t launch_cat_fold_u32(const unsigned *src, long long cells, unsigned long long *dst, hipStream_t stream);

// finalize's quad_cat lists written on the device (cat.hip): per dense pair where its table sits and
// where the live codes of column 2 (ascending key order) and the code -> key maps are in the flat arrays
struct PairListInfo { long long p_off; int kc2, o2_off, o2_len, k1_off, k2_off; };
hipError_t launch_pairlist_count(const unsigned long long *p, const int *row_pair, const int *row_code, int rows,
                                 const PairListInfo *info, const int *order_flat, unsigned *rowcnt, hipStream_t stream);
hipError_t launch_pairlist_fill(const unsigned long long *p, const int *row_pair, const int *row_code, int rows,
                                const PairListInfo *info, const int *order_flat, const int *key_flat,
                                const unsigned long long *rowbase, double *out, hipStream_t stream);

// Pair tables too big for LDS, rows binned by the high bits of code 1 (cat.hip): for every column
// col[j] that has such pairs, its partner columns part[j][0..npart) (code capacity part_kc, table at
// p + part_poff), bins of 2^shift codes (nb bins: nb << shift = the column's code capacity).
struct BinPlan {
  int ncols;
  int col[COFACTOR_MAX_CAT], shift[COFACTOR_MAX_CAT], nb[COFACTOR_MAX_CAT], npart[COFACTOR_MAX_CAT];
  int part[COFACTOR_MAX_CAT][COFACTOR_MAX_CAT], part_kc[COFACTOR_MAX_CAT][COFACTOR_MAX_CAT];
  long long part_poff[COFACTOR_MAX_CAT][COFACTOR_MAX_CAT];
};
size_t bin_scratch_words();                       // u32 words of `scratch` (zeroed once by the caller)
// codes: the piece's 16-bit code cache; binned: (1 + max npart) x out_stride u16 scratch
// d_plan: the same plan in device memory (it is past the kernel-argument size)
hipError_t launch_cat_binned_pairs(const unsigned short *codes, uint64_t rows, uint64_t stride, const BinPlan &plan,
                                   const BinPlan *d_plan, unsigned *scratch, unsigned short *binned, uint64_t out_stride,
                                   int cus, unsigned long long *p, hipStream_t stream);

// Dictionary-aligned table seam (multi-GPU): re-index tables by a code remap, and the tables as one
// array of doubles [cnt | s | p] (n_cnt + n_s + n_p values) for a single all-reduce.
hipError_t launch_cat_remap(const CatLayout &Lold, const CatDevice &Dold, const CatLayout &Lnew,
                            const CatDevice &Dnew, const int32_t *remap, hipStream_t stream);
hipError_t launch_cat_tables_export(const CatLayout &L, const CatDevice &D, double *out, hipStream_t stream);
hipError_t launch_cat_tables_import(const CatLayout &L, const CatDevice &D, const double *in, bool add,
                                    hipStream_t stream);
// the dictionary hash (host and device must agree: the host builds aligned dictionaries)
__host__ __device__ inline unsigned cat_hash_key(int32_t key, int cap) {   // cap: power of two >= 2
  int lg = 0;
  while ((1 << lg) < cap) lg++;
  return ((unsigned)key * 0x9E3779B1u) >> (32 - lg);
}

// ---- dictionary primitives shared by the kernels that work on key lists (cat.hip, ring.hip) ------------
#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long cat_pack_key(int32_t key) {
  return (1ull << 32) | (unsigned long long)(unsigned)key;
}
// inserts `key` into the column's open-addressing table (cap = power of two); flags[0] = 1 when full
__device__ __forceinline__ void cat_dict_insert(unsigned long long *slots, int cap, int32_t key, int32_t *flags) {
  const unsigned long long want = cat_pack_key(key);
  unsigned h = cat_hash_key(key, cap);
  for (int probe = 0; probe < cap; probe++) {
    const unsigned long long cur = slots[h];
    if (cur == want) return;
    if (cur == 0ull) {
      const unsigned long long old = atomicCAS(&slots[h], 0ull, want);
      if (old == 0ull || old == want) return;
    }
    h = (h + 1) & (cap - 1);
    // a table that somebody already found full is grown and the batch re-run: stop probing it (a
    // full table of `cap` slots costs `cap` dependent loads per key; 2e7 keys against a table
    // sized for 1e3 took seconds)
    if ((probe & 31) == 31 && __hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  }
  flags[0] = 1;
}
// code of a key (-1 if the dictionary does not hold it)
__device__ __forceinline__ int cat_lookup_code(const unsigned long long *slots, const int32_t *codes, int cap, int32_t key) {
  const unsigned long long want = cat_pack_key(key);
  unsigned h = cat_hash_key(key, cap);
  for (int probe = 0; probe < cap; probe++) {
    const unsigned long long cur = slots[h];
    if (cur == want) return codes[h];
    if (cur == 0ull) return -1;
    h = (h + 1) & (cap - 1);
  }
  return -1;
}
#endif

// ---- fused dense + categorical kernel for low-cardinality keys (fused.hip) ----------------------
constexpr int FUSED_MAX_SBLOCKS = 5;    // 32x32 fp32 accumulators one wave may hold (80 of its 168 registers)
// True when the shape can run on fused_kernel: triple kind, n >= 1, m >= 1, every column has at
// most 16 keys, the S accumulators fit the register budget and the LDS image fits lds_limit.
// The kernel takes whole 256-row tiles of 16-byte aligned columns (FUSED_TILE_ROWS); the caller
// sends the remaining rows (< 256) through the two-kernel path.
constexpr int FUSED_TILE_ROWS = GRAM_TILE_ROWS;
bool fused_applicable(const CatLayout &L, const int32_t *nkeys, size_t lds_limit, size_t *lds_bytes);
// workgroups the fused kernel is launched with, and the bytes of per-workgroup pair slabs it needs
int fused_grid(const CatLayout &L, int cus, int partials_cap_wgs, uint64_t rows);
size_t fused_slab_bytes(const CatLayout &L, int grid);
// mask (optional, 4-byte aligned): row filter, kept rows are added to *kept.
// skip == nullptr: every key must be in the dictionary (else flags[1]).  skip != nullptr
// (optimistic mode): skip[0] counts and skip[1..] lists the 256-row tiles that met an unknown key and
// were left out entirely; the caller redoes those tiles after a dictionary pass.
hipError_t launch_fused(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                        const CatDevice &D, int grid, double *partials, unsigned *pair_slabs,
                        unsigned *skip, double *acc, hipStream_t stream, hipEvent_t ev0 = nullptr,
                        hipEvent_t ev1 = nullptr, const uint8_t *mask = nullptr,
                        unsigned long long *kept = nullptr);
// packs the listed 256-row tiles of every column (and of the row filter, if any) into temp
hipError_t launch_gather_tiles(const NumCols &num, const CatCols &cat, int n, int m, const unsigned *list,
                               unsigned count, unsigned *temp, uint64_t temp_stride, hipStream_t stream,
                               const uint8_t *mask = nullptr, uint8_t *temp_mask = nullptr);

// ---- fused2.hip: the one-pass kernel for low-cardinality keys (<= 16 keys per column, <= 10 key
// columns, triple and NB kinds, n >= 0); whole 256-row tiles of 16-byte aligned columns ------------
bool fused2_applicable(const CatLayout &L, const int32_t *nkeys, bool masked, size_t lds_limit);
// A sub-launch of the same kernel: key counts and per-key sums of a GROUP of key columns (<= 10)
// against a GROUP of numeric columns (<= 10), nothing else (no Gram, no pair tables) — how shapes
// beyond one launch (m > 10, or per-key-sum blocks past the register budget) get their per-key
// sums off the LDS-atomic path.  Where the sub-tables sit in the aggregate's tables:
struct F2Sub {
  int s_goff[10];      // D.s index of (sub column c, code 0, numeric column 0 of the aggregate)
  int cnt_goff[10];    // D.cnt index of (sub column c, code 0)
  int n_full;          // numeric columns of the aggregate (row stride of its s tables)
  int k0;              // first numeric column of the group
  int do_cnt;          // add the key counts (one group of numeric columns does, the others do not)
};
// body rows (multiple of 256) of 16-byte aligned columns; every column of the group has <= 16 keys
// and code capacity 16.  L: the aggregate's layout; cat_idx / m_sub: the key columns of the group.
bool fused2_sub_fits(int n_sub, int m_sub, bool masked, const CatLayout &L, size_t lds_limit);
hipError_t launch_fused2_sub(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                             const CatDevice &D, int k0, int n_sub, const int *cat_idx, int m_sub, bool do_cnt,
                             int grid, size_t lds_limit, const uint8_t *mask, hipStream_t stream);
int fused2_grid(int cus, int partials_cap_wgs, uint64_t rows, int wgs_per_cu = 1);
// workgroups per CU the kernel runs with for this shape (2 where there are no pair accumulators)
int fused2_wgs_per_cu(const CatLayout &L, bool masked, size_t lds_limit);
int fused2_sub_wgs_per_cu(int n_sub, int m_sub, bool masked, const CatLayout &L, size_t lds_limit);
// rows one launch may take: the int32 pair accumulators of a wave hold 2^31 / 4096 rows
inline uint64_t fused2_max_rows(int grid) { return (uint64_t)grid * 8000ull * FUSED_TILE_ROWS; }
constexpr int FUSED2_SKIP_UNIT = 64;    // rows per entry of the optimistic pass's skip list
// skip != nullptr (optimistic mode): skip[0] counts and skip[1..] lists the 64-row blocks that met
// an unknown key and were left out entirely.
hipError_t launch_fused2(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                         const CatDevice &D, int grid, size_t lds_limit, double *partials, unsigned *pair_slabs,
                         unsigned *skip, double *acc, hipStream_t stream, hipEvent_t ev0 = nullptr,
                         hipEvent_t ev1 = nullptr, const uint8_t *mask = nullptr,
                         unsigned long long *kept = nullptr);
// ---- fused3.hip: the same one pass with specialised waves, two per SIMD (pair waves / sum waves);
// triple kind, n >= 1, 2 <= m <= 10, <= 16 keys per column.  Same contract as launch_fused2
// (skip list in FUSED2_SKIP_UNIT rows, per-workgroup pair slabs, Gram partials). -----------------------
bool fused3_applicable(const CatLayout &L, const int32_t *nkeys, bool masked, size_t lds_limit);
hipError_t launch_fused3(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L,
                         const CatDevice &D, int grid, size_t lds_limit, double *partials, unsigned *pair_slabs,
                         unsigned *skip, double *acc, hipStream_t stream, hipEvent_t ev0 = nullptr,
                         hipEvent_t ev1 = nullptr, const uint8_t *mask = nullptr,
                         unsigned long long *kept = nullptr);
// ---- nbring.hip: sum_to_nb_agg in one pass on the ring without any matrix product (NB kind, m >= 1,
// count tables + dictionaries next to the ring in LDS).  Skip list in FUSED2_SKIP_UNIT rows. ------------
bool nbring_applicable(const CatLayout &L, bool masked, size_t lds_limit);
hipError_t launch_nbring(const NumCols &num, const CatCols &cat, uint64_t rows, const CatLayout &L, const CatDevice &D,
                         int grid, size_t lds_limit, double *partials, unsigned *skip, double *acc, hipStream_t stream,
                         hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, const uint8_t *mask = nullptr,
                         unsigned long long *kept = nullptr);
// D.p[cell] += sum over the workgroups' slabs (fused2.hip)
hipError_t launch_pairs_fold2(const unsigned *slabs, int nwg, int n_p, unsigned long long *p, hipStream_t stream);
hipError_t launch_gather_units(const NumCols &num, const CatCols &cat, int n, int m, int unit, const unsigned *list,
                               unsigned count, unsigned *temp, uint64_t temp_stride, hipStream_t stream,
                               const uint8_t *mask = nullptr, uint8_t *temp_mask = nullptr);

// ---- per-row predictors (predict.hip) ------------------------------------------------------------
// out = W . [1, x_0..x_{F-1}, onehot(keys of the M key columns)] per class; argmax picks the class
// (labels == nullptr: its index), otherwise class 0's value (+ noise) is written as a float.
// mask (optional): only rows whose byte is non-zero are written.
size_t predict_lds_bytes(int F, int M, int C, int KT, size_t lds_limit, int *w_in_lds);
hipError_t launch_predict(bool argmax, const NumCols &num, const CatCols &cat, int F, int M, int C,
                          int KT, const int32_t *kbegin, const int32_t *keys, const double *W,
                          const uint8_t *mask, uint64_t rows, float *out_f, int32_t *out_i,
                          const int32_t *labels, int noise, double noise_sd,
                          unsigned long long seed, int grid, size_t lds_limit, hipStream_t stream,
                          const uint32_t *row_ids = nullptr);

}  // namespace cofactor
