// Launchers of ring.hip: batched ring operations on vectors of triples and the GROUP BY state pool.
#pragma once
#include "device.hpp"

namespace cofactor {

hipError_t launch_lift(const NumCols &num, const CatCols &cat, int n, int m, int kind, uint64_t rows,
                       const cofactor_tvec &out, hipStream_t stream);
// red: 256 doubles of scratch; acc / kept: the aggregate's accumulator image and kept-row counter
hipError_t launch_tvec_dense(const cofactor_tvec &v, double *red, double *acc, unsigned long long *kept, int grid,
                             hipStream_t stream);
// pass 0: keys into the dictionaries; pass 1: values into the count / sum / pair tables
hipError_t launch_tvec_keys(const cofactor_tvec &v, const CatLayout &L, const CatDevice &D, int pass, hipStream_t stream);
// multiply_triple (mulfill.hip): per-row totals of the three list families; the one-pass fill at the rows' places
// quad_cat entries of the pairs L keeps as sorted lists: fill == 0 counts them per pair (counts, zeroed by the
// caller); fill == 1 writes them as (packed key pair, count) from base[pair] on (counts zeroed again: cursors)
hipError_t launch_tvec_sparse(const cofactor_tvec &v, const CatLayout &L, unsigned long long *counts,
                              const unsigned long long *base, unsigned long long *keys, unsigned long long *cnt, int fill,
                              hipStream_t stream);
hipError_t launch_mul_pair_lens(const cofactor_tvec &a, const uint32_t *asel, const cofactor_tvec &b, const uint32_t *bsel,
                                uint64_t rows, uint64_t *t0, uint64_t *t1, uint64_t *t2, hipStream_t stream);
hipError_t launch_mul_fill(const cofactor_tvec &a, const uint32_t *asel, const cofactor_tvec &b, const uint32_t *bsel,
                           uint64_t rows, const uint64_t *base0, const uint64_t *base1, const uint64_t *base2,
                           const cofactor_tvec &out, uint64_t entries, int cus, hipStream_t stream);
hipError_t ring_exclusive_scan(const uint64_t *len, uint64_t *offs, uint64_t items, void *temp, size_t *temp_bytes,
                               hipStream_t stream);

hipError_t launch_groups_insert(const int32_t *gid, const CatCols &cat, uint64_t rows, const CatLayout &L,
                                const CatDevice &D, const CatLayout &Lg, const CatDevice &Dg, int is_key, hipStream_t stream);
hipError_t launch_max_i32(const int32_t *v, uint64_t rows, int *out, hipStream_t stream);
hipError_t launch_groups_accumulate(const int32_t *gid, const NumCols &num, const CatCols &cat, uint64_t rows,
                                    const CatLayout &L, const CatDevice &D, const CatLayout &Lg, const CatDevice &Dg,
                                    int is_key, double *tab, long long dtot, int grid, hipStream_t stream);
// segmented path (groupseg.hip): numeric-only triples; rows <= 2^27 per call; scratch of groups_seg_scratch_bytes.
// miss: device word set when a row's group is unknown; phase 0: the whole sequence, 2: the counting
// pass only, 1: the rest after a phase-2 call on the same arguments.
int groups_seg_record_floats(int n);
size_t groups_seg_scratch_bytes(int n, uint64_t rows, long long groups, int cus);
hipError_t launch_groups_segmented(const int32_t *gid, const NumCols &num, int n, uint64_t rows, const CatLayout &Lg,
                                   const CatDevice &Dg, int is_key, long long groups, double *tab, long long dtot,
                                   void *scratch, int cus, int32_t *miss, int phase, hipStream_t stream);
hipError_t launch_groups_relayout(const CatLayout &Lo, const CatLayout &Ln, const double *to, double *tn, long long dto,
                                  long long dtn, long long groups, hipStream_t stream);
hipError_t launch_groups_combine(double *tab, long long dtot, long long dst, long long src, hipStream_t stream);
hipError_t launch_groups_lists(const double *tab, long long dtot, const CatLayout &L, const int32_t *gorder, long long groups,
                               const int32_t *ord, const int32_t *keyof, int family, uint64_t *len, const uint64_t *offs,
                               const cofactor_tvec &out, int mode, hipStream_t stream);
hipError_t launch_groups_dense(const double *tab, long long dtot, int n, int T, const int32_t *gorder, long long groups,
                               const cofactor_tvec &out, hipStream_t stream);

}  // namespace cofactor
