/*
 * cofactor_hip.h — C ABI of libcofactor_hip.so: the MI355X (gfx950) implementation of the
 * cofactor-triple ("ring") aggregate hot path of eddbase/duckdb-imputation.
 *
 * This header is the drop-in boundary.  Every entry point names the reference
 * interface it stands in for (paths relative to the reference repo root).  Plain pointers and
 * sizes only; no C++ / torch / DuckDB types.  No exceptions cross it: every call returns a
 * cofactor_status and cofactor_last_error() gives the thread-local message.
 *
 * Threading contract (mirrors DuckDB's: duckdb_extension/src/triple/sum/sum_state.cpp has no
 * locks): distinct cofactor_agg handles may be used from different threads concurrently; one
 * handle is used by one thread at a time.  Aggregates of one context share its stream and scratch
 * buffers; the library serialises their device work with a per-context lock (host-side staging
 * of different aggregates still runs in parallel).
 *
 * ---------------------------------------------------------------------------------------------
 * Flat triple blob — the host representation of ONE finalised triple: an array of doubles (all
 * int32 keys / counts / float sums are exactly representable), in the order in which the
 * reference's finalize fills its nested vectors (sum_state.cpp:116-464):
 *
 *   [0] kind (0 = COFACTOR_TRIPLE, 1 = COFACTOR_NB)     [1] n (#numeric)     [2] m (#categorical)
 *   [3] N
 *   lin[n]
 *   quad[n(n+1)/2]  row-major upper triangle (kind 0)    |    quad[n] diagonal (kind 1)
 *   lin_cat:       m lists, each:  len, len x (key, count)              keys ascending
 *   quad_num_cat:  n*m lists, index k*m + c, each: len, len x (key, sum of x_k)   (kind 0 only)
 *   quad_cat:      m(m+1)/2 lists (c1 outer, c2 >= c1), each: len, len x (key1, key2, count),
 *                  lexicographically ascending                                     (kind 0 only)
 *
 * Sums are carried in double; a DuckDB FLOAT field is the value rounded to float.
 * ---------------------------------------------------------------------------------------------
 */
#ifndef COFACTOR_HIP_H
#define COFACTOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: only what this header declares is exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define COFACTOR_ABI_VERSION 3
/* sum_to_triple_<x>_<y> is registered for x,y in 0..20 (the reference registers 0..19,
 * duckdb_imputation_extension.cpp:80-84; README.md:136 documents "up to 20"). */
#define COFACTOR_MAX_NUM 20
#define COFACTOR_MAX_CAT 20

typedef enum {
  COFACTOR_OK = 0,
  COFACTOR_ERR_INVALID = 1,     /* bad argument (null pointer, n/m out of range, kind mismatch)  */
  COFACTOR_ERR_NO_DEVICE = 2,   /* no usable gfx950 device / HIP runtime error at context create */
  COFACTOR_ERR_HIP = 3,         /* a HIP call failed; message carries hipGetErrorString           */
  COFACTOR_ERR_CAPACITY = 4,    /* output buffer too small; *needed tells how many doubles        */
  COFACTOR_ERR_UNSUPPORTED = 5, /* shape outside what the device path handles (message says which)*/
  COFACTOR_ERR_INTERNAL = 6     /* a device-side invariant broke (a row met a key its dictionary
                                   pass had not registered); reported by the first call that
                                   synchronises with the update that caused it                   */
} cofactor_status;

typedef enum { COFACTOR_TRIPLE = 0, COFACTOR_NB = 1 } cofactor_kind;

typedef struct cofactor_ctx cofactor_ctx; /* one GPU: device id, stream, workspaces            */
typedef struct cofactor_agg cofactor_agg; /* one aggregate state (what a Triple::SumState is)   */

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *cofactor_last_error(void);
int cofactor_abi_version(void);

/* Number of GPUs HIP shows to this process (0 when there is none): what a host that opens one
 * context per GPU iterates over. */
int cofactor_device_count(void);

/* ---- context ------------------------------------------------------------------------------ */
/* Opens HIP device `device`.  Fails with COFACTOR_ERR_NO_DEVICE when there is no GPU: there is
 * no CPU fallback in this library. */
cofactor_status cofactor_ctx_create(int device, cofactor_ctx **out);
void cofactor_ctx_destroy(cofactor_ctx *ctx);
cofactor_status cofactor_ctx_synchronize(cofactor_ctx *ctx);
/* The hipStream_t all of this context's kernels are launched on (for event timing). */
void *cofactor_ctx_stream(cofactor_ctx *ctx);

/* Optional timing of the streaming kernels (gram_kernel, cat_accumulate_kernel, fused_kernel) with
 * HIP events recorded on the context stream right around each launch (bench.py's roofline
 * figures).  read synchronises the stream, returns the summed kernel milliseconds and launch
 * counts since the previous read, and clears them.  Any output pointer may be NULL. */
cofactor_status cofactor_ctx_profile_enable(cofactor_ctx *ctx, int on);
/* Name of the one-pass kernel the context launched last ("fused_kernel", "fused2_kernel",
 * "fused3_kernel"; "" if none yet): what the `fused` figures of profile_read belong to. */
const char *cofactor_ctx_profile_kernel(cofactor_ctx *ctx);
cofactor_status cofactor_ctx_profile_read(cofactor_ctx *ctx, double *gram_ms,
                                          uint64_t *gram_launches, double *cat_ms,
                                          uint64_t *cat_launches, double *fused_ms,
                                          uint64_t *fused_launches);

/* Measurement support (bench.py's calibrated roofline, SURVEY.md §8d "report both"): GB/s a plain
 * float4 streaming kernel reaches on this GPU over a scratch buffer of `bytes` bytes, `reps`
 * launches between HIP events: copy (read + written bytes counted) and read-only. */
cofactor_status cofactor_ctx_calibrate(cofactor_ctx *ctx, uint64_t bytes, int reps, double *copy_gbs,
                                       double *read_gbs);

/* ---- aggregate state ------------------------------------------------------------------------
 * Replaces Triple::SumState + StateFunction::Initialize/Destroy
 * (duckdb_extension/src/include/triple/sum/sum_state.h:14-57).  n/m are what the reference
 * derives from the argument types on the first update (sum_no_lift.cpp:66-73,96-99). */
cofactor_status cofactor_agg_create(cofactor_ctx *ctx, int n_num, int n_cat, cofactor_kind kind,
                                    cofactor_agg **out);
void cofactor_agg_destroy(cofactor_agg *agg);
/* Back to the freshly created state (N = 0, all tables empty, dictionaries kept allocated). */
cofactor_status cofactor_agg_reset(cofactor_agg *agg);

/* update — Triple::SumNoLift (duckdb_extension/src/triple/sum/sum_no_lift.cpp:53-216) and
 * Triple::sum_to_nb_agg (duckdb_extension/src/triple/sum/sum_to_nb_agg.cpp:39-146).
 *
 * Cardinalities: any number of distinct keys per column up to 2^27 (the per-key count / sum tables
 * are dense in the key's code).  A pair table (sum_no_lift.cpp:195-214) is dense, code-indexed,
 * while it has at most 2^26 cells and all dense pair tables together at most 2^30; beyond that the
 * pair is kept as a sorted (key1, key2) -> count list in device memory, like the reference's
 * std::map.  States that hold such lists work with update / sum_triple / combine / finalize / reset
 * and with the multi-GPU seam (the dense tables are aligned and all-reduced as usual, the lists are
 * gathered and merged: cofactor_agg_sparse_*, cofactor_agg_allreduce).  The GROUP BY pool
 * (cofactor_groups) keeps dense per-group tables only and refuses cardinalities beyond them.
 *
 * Device form: the columns are resident in this context's HBM (d_num[k] -> float[rows],
 * d_cat[c] -> int32[rows]; the pointer arrays themselves are host arrays).  Asynchronous on the
 * context stream: work the caller queued on OTHER streams that writes these columns must have
 * completed (or be ordered before the context stream, cofactor_ctx_stream) when this is called,
 * and the columns must stay untouched until the stream has passed the update
 * (cofactor_ctx_synchronize, or any finalize).  This is the measured hot path. */
cofactor_status cofactor_agg_update_device(cofactor_agg *agg, const float *const *d_num,
                                           const int32_t *const *d_cat, uint64_t rows);

/* Device form with a row filter: d_mask[i] != 0 keeps row i, 0 drops it (one byte per row, device
 * memory).  This is the aggregate under a WHERE clause evaluated on the GPU — the MICE drivers'
 *   SELECT sum_to_triple_n_m(...) FROM t WHERE <col>_IS_NULL IS FALSE
 * (imputation/algorithms/imputation_base.cpp:21-34, 92-100), where DuckDB would hand update a
 * selection vector.  N counts kept rows only. */
cofactor_status cofactor_agg_update_device_masked(cofactor_agg *agg, const float *const *d_num,
                                                  const int32_t *const *d_cat,
                                                  const uint8_t *d_mask, uint64_t rows);

/* Host form: what DuckDB hands the aggregate's update callback — one DataChunk (<= 2048 rows in
 * DuckDB, any size here) of host columns in UnifiedVectorFormat.  Column k's value for logical
 * row i is  col[k][ sel && sel[k] ? sel[k][r] : r ]  with  r = row_idx ? row_idx[i] : i
 * (input_data[k].sel->get_index(i), sum_no_lift.cpp:120; row_idx lists the chunk rows whose
 * state pointer is this state, sum_no_lift.cpp:84,94,139).  num_sel / cat_sel / row_idx may be
 * NULL.  Rows are staged in pinned memory and flushed to the GPU in large batches. */
cofactor_status cofactor_agg_update_host(cofactor_agg *agg, const float *const *num,
                                         const int32_t *const *cat,
                                         const uint32_t *const *num_sel,
                                         const uint32_t *const *cat_sel, const uint32_t *row_idx,
                                         uint64_t rows);

/* update with already-lifted triples — Triple::Sum (duckdb_extension/src/triple/sum/sum.cpp:
 * 57-261) and Triple::sum_nb_agg (sum/sum_nb_agg.cpp:45-175): `count` blobs, concatenated,
 * blob i at blobs[offsets[i] .. offsets[i+1]). */
cofactor_status cofactor_agg_update_triples(cofactor_agg *agg, const double *blobs,
                                            const uint64_t *offsets, uint64_t count);

/* combine — Triple::SumStateCombine (duckdb_extension/src/triple/sum/sum_state.cpp:10-114):
 * dst += src, on the device: the accumulator image is added by a kernel (sum_state.cpp:25,73-83); the
 * two states' dictionaries are aligned to the union of their key lists (only the key lists, a few
 * KB, pass through the host, and nothing at all while both states still hold a common alignment),
 * then src's table image [cnt | s | p] is added into dst's (sum_state.cpp:87-111 as a dense sum) and
 * sorted pair lists are merged.  src keeps its value (its tables may be re-indexed).  The handles may
 * live on different contexts, also on different GPUs: the image then travels by a peer copy (xGMI)
 * — this is how a DuckDB process with one context per GPU merges its thread-local states. */
cofactor_status cofactor_agg_combine(cofactor_agg *dst, cofactor_agg *src);

/* finalize — Triple::SumStateFinalize (sum_state.cpp:116-464): writes the flat triple blob.
 * Two-call protocol: with out == NULL or cap too small returns COFACTOR_ERR_CAPACITY (or OK if
 * out == NULL) and sets *needed.  Synchronises the context stream. */
cofactor_status cofactor_agg_finalize(cofactor_agg *agg, double *out, uint64_t cap,
                                      uint64_t *needed);

/* ---- multi-GPU seam -----------------------------------------------------------
 * The dense part of the partial triple as ONE device array of doubles
 *   [ N, lin[n], quad[n(n+1)/2 | n] ]            (cofactor_dense_len(n, kind) values)
 * so that a single RCCL all-reduce(sum) over the ranks' arrays is the dense half of
 * SumStateCombine (sum_state.cpp:25,73-83).  Both calls only ENQUEUE a small kernel on the
 * context stream (cofactor_ctx_stream) and return; nothing is copied to the host.
 *   export: d_out (device memory, caller-owned, e.g. a torch tensor) receives the state's
 *           current totals once the stream has passed the call; the collective must be ordered
 *           after the context stream (run it on that stream, or wait for it).
 *   import: the state's dense totals BECOME the values in d_in (after the all-reduce).  d_in must
 *           be complete when the context stream reaches the call and stay valid until it has
 *           passed it.  Categorical tables are not touched. */
uint64_t cofactor_dense_len(int n_num, cofactor_kind kind);
cofactor_status cofactor_agg_export_dense_device(cofactor_agg *agg, double *d_out);
cofactor_status cofactor_agg_import_dense_device(cofactor_agg *agg, const double *d_in);

/* The categorical half of SumStateCombine across ranks (sum_state.cpp:87-111: merge of the
 * per-key maps) as dense, dictionary-aligned tables:
 *   1. every rank lists its keys (cofactor_agg_keys), the lists are all-gathered;
 *   2. every rank calls cofactor_agg_align_keys with the SAME concatenation of all lists: the
 *      state's dictionaries become "code = rank of the key in the sorted union" and its count /
 *      per-key-sum / pair-count tables are re-indexed on the device, so that all ranks now hold
 *      tables of identical shape and meaning (values the state held on the host under keys are
 *      folded into the tables);
 *   3. ONE all-reduce(sum) over the table image [cnt | s | p] (cofactor_agg_tables_len doubles;
 *      counts are exact integers in doubles), exported and imported on the device like the
 *      dense part, is the merge.
 * cofactor_agg_dict_signature tells whether step 1-2 can be skipped: it is non-zero and equal on
 * all ranks exactly when every rank's dictionaries are still the ones of the last common
 * alignment (no rank met a new key since).
 *
 * keys: two-call protocol (int32 entries); offsets[m+1] delimit the columns, keys ascending.
 * align_keys: keys[offsets[c] .. offsets[c+1]) = any superset of the state's keys of column c, in
 * any order, duplicates allowed. */
cofactor_status cofactor_agg_keys(cofactor_agg *agg, int32_t *out, uint64_t cap, uint64_t *needed,
                                  uint64_t *offsets);
cofactor_status cofactor_agg_dict_signature(cofactor_agg *agg, uint64_t *sig);
cofactor_status cofactor_agg_align_keys(cofactor_agg *agg, const int32_t *keys,
                                        const uint64_t *offsets);
uint64_t cofactor_agg_tables_len(cofactor_agg *agg);
cofactor_status cofactor_agg_export_tables_device(cofactor_agg *agg, double *d_out);
cofactor_status cofactor_agg_import_tables_device(cofactor_agg *agg, const double *d_in);

/* Pair tables kept as sorted lists (see update) across ranks: list q (upper-triangle index, c1 outer,
 * c2 >= c1) is `lens[q]` entries of (packed key pair, count), packed = (key1 ^ 2^31) << 32 | (key2 ^
 * 2^31), ascending.  The ranks gather each other's lists and every rank hands the concatenation of
 * ALL ranks' lists (its own included, any order, duplicates allowed) to sparse_assign, which sorts,
 * adds up equal keys and makes the result the state's list.  is_list tells whether pair q is a list
 * under the state's current (aligned) layout — the same answer on every rank after align_keys. */
cofactor_status cofactor_agg_sparse_lens(cofactor_agg *agg, uint64_t *lens, uint64_t cap);
cofactor_status cofactor_agg_sparse_is_list(cofactor_agg *agg, int32_t pair, int32_t *is_list);
cofactor_status cofactor_agg_sparse_export_device(cofactor_agg *agg, int32_t pair, uint64_t *d_keys,
                                                  uint64_t *d_counts);
cofactor_status cofactor_agg_sparse_assign_device(cofactor_agg *agg, int32_t pair, const uint64_t *d_keys,
                                                  const uint64_t *d_counts, uint64_t len);

/* The whole seam below the C ABI: a communicator of the library itself (RCCL, loaded with dlopen when
 * the first communicator is made; no torch, no MPI) and ONE call that turns every rank's state into
 * the merge of all ranks' states — Triple::SumStateCombine across GPUs (sum_state.cpp:10-114):
 *   a small all-gather of (status, dictionary signature, key counts); only when some rank's
 *   dictionaries changed since the last common alignment, an all-gather of the key lists and
 *   cofactor_agg_align_keys; ONE ncclAllReduce(sum, double) of [N, lin, quad | cnt | s | p] on the
 *   context stream between the export and import kernels; sorted pair lists gathered and merged.
 * Every rank returns the same status: a rank that fails before the all-reduce says so in the first
 * exchange and all ranks return COFACTOR_ERR_INVALID together instead of hanging in the collective.
 * One process per GPU: rank 0 makes the 128-byte id, the caller ships it to the other ranks (the
 * launcher's rendezvous, a file, a socket), every rank calls comm_create with the same id. */
typedef struct cofactor_comm cofactor_comm;
#define COFACTOR_COMM_ID_BYTES 128
cofactor_status cofactor_comm_unique_id(void *id_out);
cofactor_status cofactor_comm_create(cofactor_ctx *ctx, const void *id, int rank, int world,
                                     cofactor_comm **out);
void cofactor_comm_destroy(cofactor_comm *comm);
cofactor_status cofactor_agg_allreduce(cofactor_agg *agg, cofactor_comm *comm);

/* ---- scalar ring ops on flat triple blobs (host; tiny per-row work) ---------------------------
 * All use the two-call protocol of cofactor_agg_finalize. */

/* to_cofactor / to_nb_agg — Triple::CustomLift (duckdb_extension/src/triple/lift.cpp:15-243),
 * Triple::to_nb_lift (triple/lift_to_nb_agg.cpp:13-136): one blob per row, concatenated;
 * offsets[rows+1] (may be NULL).  Host columns with the same sel semantics as update_host. */
cofactor_status cofactor_lift_host(const float *const *num, int n_num, const int32_t *const *cat,
                                   int n_cat, uint64_t rows, cofactor_kind kind, double *out,
                                   uint64_t cap, uint64_t *needed, uint64_t *offsets);

/* multiply_triple / multiply_nb_agg — Triple::MultiplyFunction (triple/mul.cpp:19-611),
 * Triple::multiply_nb (triple/mul_nb.cpp:20-268). */
cofactor_status cofactor_triple_multiply(const double *a, uint64_t a_len, const double *b,
                                         uint64_t b_len, double *out, uint64_t cap,
                                         uint64_t *needed);

/* Value-level t1 + t2 / t1 - t2 — Triple::sum_triple (imputation/triple/sum.cpp:68-209; also
 * duckdb_extension/src/triple/sum/sum.cpp:319-460), Triple::sum_nb_triple
 * (imputation/triple/sum_nb.cpp:38-83), Triple::subtract_triple (imputation/triple/sub.cpp:
 * 71-217).  On subtract a key missing from `a` is reported via cofactor_last_error() and
 * skipped, as the reference prints and skips (sub.cpp:28-29). */
cofactor_status cofactor_triple_add(const double *a, uint64_t a_len, const double *b,
                                    uint64_t b_len, double *out, uint64_t cap, uint64_t *needed);
cofactor_status cofactor_triple_sub(const double *a, uint64_t a_len, const double *b,
                                    uint64_t b_len, double *out, uint64_t cap, uint64_t *needed);

/* Number of doubles in the blob starting at `blob` (walks the lists); 0 if the blob is malformed
 * or does not end within `cap` doubles — nothing at or beyond blob[cap] is read.  Every entry
 * point that takes a blob takes its extent (a_len, b_len, triple_len, offsets[i+1]) and checks
 * every list header against it before reading. */
uint64_t cofactor_blob_len(const double *blob, uint64_t cap);

/* Text round trip.  The reference's MICE drivers carry a triple from the aggregate
 * to the trainer as TEXT: Value::ToString() of the result, pasted into the next SQL statement and
 * cast back (imputation/algorithms/imputation_base.cpp:46-49,116).  to_text writes DuckDB's STRUCT
 * literal ({'N': 5, 'lin_agg': [15.0, ..], 'lin_cat': [[{'key': 4, 'value': 3.0}, ..], ..], ..}) with
 * the shortest digits that read back as the same value; from_text parses it (either field-name
 * flavour, any whitespace).  blob -> text -> blob is the identity.  Two-call protocol; `needed`
 * counts bytes including the terminating NUL for to_text, doubles for from_text. */
cofactor_status cofactor_triple_to_text(const double *blob, uint64_t blob_len, int32_t aggregate_names,
                                        char *out, uint64_t cap, uint64_t *needed);
cofactor_status cofactor_triple_from_text(const char *text, uint64_t text_len, double *out,
                                          uint64_t cap, uint64_t *needed);

/* ---- batched ring ops on the GPU (the factorised-join pipeline) ----------------
 *
 *   SELECT sum_triple(multiply_triple(A, B)) FROM
 *     (SELECT gb, sum_to_triple_2_2(b,c,d,e) AS A FROM test1 GROUP BY gb) a  JOIN
 *     (SELECT gb, sum_to_triple_2_2(a,c,d,f) AS B FROM test2 GROUP BY gb) b  ON a.gb = b.gb
 *                                                               (the reference's README.md:163-173)
 *
 * cofactor_tvec — a VECTOR of triples in device (or host) memory, laid out exactly as the
 * reference's result STRUCT vector is laid out by DuckDB after RecursiveFlatten
 * (sum_state.cpp:116-464, lift.cpp:225-241): one array per leaf field, list_entry_t =
 * {uint64 offset, uint64 length} pairs for every list level.  Row i of the vector:
 *   N[i];  lin[lin_e[i]];  quad[quad_e[i]]                       (FLOAT children of two LISTs)
 *   lin_cat:      outer entry lc_outer[i] -> m sub-lists lc_sub[..] -> entries (lc_key, lc_val)
 *   quad_num_cat: outer entry nc_outer[i] -> n*m sub-lists (index k*m + c) -> (nc_key, nc_val)
 *   quad_cat:     outer entry cc_outer[i] -> m(m+1)/2 sub-lists -> (cc_key1, cc_key2, cc_val)
 * Every row has the same n and m (as the reference assumes, sum.cpp:99-106, mul.cpp:57-69).
 * For kind = COFACTOR_NB quad holds the n diagonal entries and the nc / cc members are unused.
 * `list_entry` arrays are pairs of uint64 (offset, length), i.e. a duckdb::list_entry_t array can
 * be passed as is. */
typedef struct {
  uint64_t count;                 /* rows (triples) */
  int32_t n, m, kind;
  int32_t *N;                     /* [count] */
  uint64_t *lin_e;  float *lin;   /* [count] entries; lin child [count * n] */
  uint64_t *quad_e; float *quad;  /* [count] entries; quad child */
  uint64_t *lc_outer, *lc_sub; int32_t *lc_key; float *lc_val;
  uint64_t *nc_outer, *nc_sub; int32_t *nc_key; float *nc_val;
  uint64_t *cc_outer, *cc_sub; int32_t *cc_key1, *cc_key2; float *cc_val;
  /* entries of the payload arrays: capacity for the ops that write them, extent for inputs */
  uint64_t lc_cap, nc_cap, cc_cap;
  /* extents of the other arrays (only the *_host entry points, which copy them, look at these):
   * floats in lin / quad, (offset, length) pairs in lc_sub / nc_sub / cc_sub */
  uint64_t lin_len, quad_len, lc_subs, nc_subs, cc_subs;
} cofactor_tvec;

/* to_cofactor / to_nb_agg on the GPU — Triple::CustomLift (triple/lift.cpp:15-243), to_nb_lift
 * (triple/lift_to_nb_agg.cpp:13-136): one triple per row of the device columns, written into the
 * caller's device arrays `out` (regular shape: every sub-list has exactly one entry, so
 * lc_cap >= rows*m, nc_cap >= rows*n*m, cc_cap >= rows*m(m+1)/2; lin [rows*n], quad
 * [rows*n(n+1)/2 | rows*n]).  An expand kernel, HBM-write bound:
 *   4 (1 + n + T) + 32 + 16 (3 + m + n m + T_m) + 8 (m + n m) + 12 T_m   bytes written per row
 * against 4 (n + m) read  (T = n(n+1)/2 or n, T_m = m(m+1)/2; 0 for the nc / cc parts of NB). */
cofactor_status cofactor_lift_device(cofactor_ctx *ctx, const float *const *d_num, int n_num,
                                     const int32_t *const *d_cat, int n_cat, uint64_t rows,
                                     cofactor_kind kind, cofactor_tvec *out);

/* sum_triple / sum_nb_agg on the GPU — Triple::Sum (triple/sum/sum.cpp:57-261), sum_nb_agg
 * (sum_nb_agg.cpp:45-175): adds every row of the device vector `v` to the state.  The dense
 * children are reduced column-wise by a streaming kernel (4 (1 + n + T) bytes read per row); the
 * key lists go through the state's dictionaries into its count / sum / pair tables. */
cofactor_status cofactor_agg_update_tvec_device(cofactor_agg *agg, const cofactor_tvec *v);

/* multiply_triple / multiply_nb_agg on the GPU — Triple::MultiplyFunction (triple/mul.cpp:19-611),
 * multiply_nb (mul_nb.cpp:20-268): out row i = a row (a_sel ? a_sel[i] : i)  x  b row
 * (b_sel ? b_sel[i] : i), i < rows (the selection vectors are what a join hands the function).
 * Two-call protocol on the payload sizes: with out == NULL (or capacities too small ->
 * COFACTOR_ERR_CAPACITY) only *lc_need / *nc_need / *cc_need are set (entries of the three payload
 * arrays).  N is the int32 product, as in the reference (mul.cpp:46-49).  The size query
 * synchronises (it returns numbers); the fill is asynchronous on the context stream like the other
 * device entry points (cofactor_ctx_synchronize before another stream reads `out`).
 * The size query leaves its plan (the rows' payload sizes and places) with the context; the fill call
 * that FOLLOWS it with the same a / b / selection vectors / rows takes the plan over instead of
 * computing it again — the inputs must not be modified between the two calls.  A fill call without
 * such a query (capacities known to suffice) computes the plan itself: one call, one synchronisation.
 * GROUP BY pool, numeric-only triples: see cofactor_groups_update_device below. */
cofactor_status cofactor_multiply_device(cofactor_ctx *ctx, const cofactor_tvec *a,
                                         const uint32_t *d_a_sel, const cofactor_tvec *b,
                                         const uint32_t *d_b_sel, uint64_t rows, cofactor_tvec *out,
                                         uint64_t *lc_need, uint64_t *nc_need, uint64_t *cc_need);

/* The same three over HOST arrays (a DuckDB DataChunk / result vector): staged to the device,
 * computed there, copied back.  `out` arrays are host memory sized by the caller (lift: regular
 * shape, see above; multiply: two-call protocol). */
cofactor_status cofactor_lift_host_tvec(cofactor_ctx *ctx, const float *const *num, int n_num,
                                        const int32_t *const *cat, int n_cat, uint64_t rows,
                                        cofactor_kind kind, cofactor_tvec *out);
cofactor_status cofactor_agg_update_tvec_host(cofactor_agg *agg, const cofactor_tvec *v);
cofactor_status cofactor_multiply_host(cofactor_ctx *ctx, const cofactor_tvec *a, const uint32_t *a_sel,
                                       const cofactor_tvec *b, const uint32_t *b_sel, uint64_t rows,
                                       cofactor_tvec *out, uint64_t *lc_need, uint64_t *nc_need,
                                       uint64_t *cc_need);

/* GROUP BY: all groups of one aggregate in ONE device state pool, one kernel launch per batch
 * whatever the number of groups — the per-row state pointers of Triple::SumNoLift
 * (sum_no_lift.cpp:84,94,139,161,201) as a per-row group id.  Group g owns row g of a dense
 * table [ N | lin | quad | key counts | per-key sums | pair counts ] (dictionary-coded keys), so a
 * join key with 1e5 values costs 1e5 table rows, not 1e5 states with their own buffers.
 *   gid: group of every row — slot ids 0..G-1 handed out by the caller (is_key = 0; the DuckDB
 *        glue numbers its SumStates), or arbitrary int32 keys (is_key = 1; GROUP BY column).
 * update:  numeric-only triples (n_cat = 0) with many rows per group in the batch regroup the batch by
 *          group and run one matrix-core Gram per group (csrc/groupseg.hip, scratch of 132 B per row
 *          at 20_0 kept by the context); everything else adds cell by cell with fp64 atomics.
 *          Host batches (update_host) are collected in a pinned staging block and reach the device 2^18
 *          rows at a time, or when another entry point looks at the pool (count, combine, finalize,
 *          to_tvec, update_device); a device error of staged rows is reported by that later call.
 * combine: Triple::SumStateCombine per group (dst += src; src unchanged).
 * finalize: one group's triple as a flat blob (two-call protocol), keys ascending.
 * to_tvec: every group's triple, in ascending group order (is_key = 1: ascending key; the keys go to
 *          d_group_keys if not NULL), as one device vector — the input of cofactor_multiply_device.
 *          Two-call protocol like cofactor_multiply_device. */
typedef struct cofactor_groups cofactor_groups;
cofactor_status cofactor_groups_create(cofactor_ctx *ctx, int n_num, int n_cat, cofactor_kind kind,
                                       int is_key, cofactor_groups **out);
void cofactor_groups_destroy(cofactor_groups *grp);
cofactor_status cofactor_groups_update_device(cofactor_groups *grp, const int32_t *d_gid,
                                              const float *const *d_num,
                                              const int32_t *const *d_cat, uint64_t rows);
cofactor_status cofactor_groups_update_host(cofactor_groups *grp, const int32_t *gid,
                                            const float *const *num, const int32_t *const *cat,
                                            uint64_t rows);
cofactor_status cofactor_groups_count(cofactor_groups *grp, uint64_t *n_groups);
cofactor_status cofactor_groups_combine(cofactor_groups *grp, int32_t dst_gid, int32_t src_gid);
/* slot-id pools (is_key = 0): group `gid` back to the empty triple, so that the slot of a state DuckDB
 * has destroyed (after its combine / finalize) can serve the next new group — without it a prepared
 * statement executed again and again only ever grows its table. */
cofactor_status cofactor_groups_reset_group(cofactor_groups *grp, int32_t gid);
cofactor_status cofactor_groups_finalize(cofactor_groups *grp, int32_t gid, double *out,
                                         uint64_t cap, uint64_t *needed);
cofactor_status cofactor_groups_to_tvec(cofactor_groups *grp, cofactor_tvec *out,
                                        int32_t *d_group_keys, uint64_t *lc_need,
                                        uint64_t *nc_need, uint64_t *cc_need);

/* ---- consumers of the triple (what one MICE iteration needs) --------------
 * Training is host fp64 over the p x p cofactor matrix (p = 1 + n + #keys, independent of the
 * row count) and emits the reference's flat FLOAT[] parameter vector; prediction is a HIP kernel
 * over device-resident columns.  `out`/`cap`/`needed` count floats, two-call protocol as above. */

/* linreg_train — ML::ridge_linear_regression (duckdb_extension/src/ML/regression.cpp:108-354):
 * gradient descent with Barzilai-Borwein steps and backtracking on the one-hot sigma matrix
 * (ML/utils.cpp:176-310).  label = index of the numeric column to predict (0-based).  Output
 * [m, begin[0..m], keys.., intercept, coefficients of the other numeric columns, of every key,
 *  (their means if normalize), (residual std if compute_variance)]  (regression.cpp:313-353). */
cofactor_status cofactor_linreg_train(const double *triple, uint64_t triple_len, int32_t label,
                                      float step_size,
                                      float lambda, int32_t max_iterations,
                                      int32_t compute_variance, int32_t normalize, float *out,
                                      uint64_t cap, uint64_t *needed);

/* lda_train — lda_train (duckdb_extension/src/ML/lda.cpp:161-416): shrinkage LDA, class means and
 * pooled within-class covariance from the triple, solved by minimum-norm least squares
 * (dgelsd there).  label = index of the key column holding the class.  Output
 * [C, #idx, begin offsets of the other key columns.., their keys.., class keys[C],
 *  coef[C][p], intercept[C], (means[p] if normalize)]  (lda.cpp:335-386). */
cofactor_status cofactor_lda_train(const double *triple, uint64_t triple_len, int32_t label,
                                   float shrinkage,
                                   int32_t normalize, float *out, uint64_t cap, uint64_t *needed);

/* linreg_predict — ML::linreg_impute (regression.cpp:397-508): per row intercept + coef . x +
 * the coefficient of each key column's key (a key the model never saw adds 0), optionally plus
 * N(0, params[last]) noise.  The reference draws the noise from random() seeded off
 * /dev/urandom; here it is a counter-based generator of (seed, row): row i of the call draws
 * hash(seed + 0x9E3779B97F4A7C15 * (i + 1)).  A run is reproducible, and independent of the sharding
 * when a rank whose columns start at row `first` of the whole table passes
 * seed + 0x9E3779B97F4A7C15 * first (mod 2^64), as cofactor_hip/mice.py does.  d_num[n_num] are the feature columns in training order WITHOUT
 * the label, d_cat[n_cat] every key column.  d_mask (optional, one byte per row): only rows with
 * a non-zero byte are written — `CASE WHEN col_IS_NULL THEN linreg_predict(..) ELSE col END`
 * (imputation/algorithms/imputation_base.cpp:137) as an in-place column update. */
cofactor_status cofactor_linreg_predict_device(cofactor_ctx *ctx, const float *params,
                                               uint64_t n_params, int32_t noise,
                                               int32_t normalize, uint64_t seed,
                                               const float *const *d_num, int32_t n_num,
                                               const int32_t *const *d_cat, int32_t n_cat,
                                               const uint8_t *d_mask, uint64_t rows, float *d_out);

/* The same over rows that have been REORDERED (a table partitioned by its null pattern,
 * imputation/algorithms/imputation_low.cpp): d_row_ids[i] (optional) is the index row i had in the
 * original table — it draws the noise of that place, hash(seed + 0x9E3779B97F4A7C15 * (d_row_ids[i] + 1)),
 * so the imputed values do not depend on the reordering. */
cofactor_status cofactor_linreg_predict_rows_device(cofactor_ctx *ctx, const float *params,
                                                    uint64_t n_params, int32_t noise,
                                                    int32_t normalize, uint64_t seed,
                                                    const float *const *d_num, int32_t n_num,
                                                    const int32_t *const *d_cat, int32_t n_cat,
                                                    const uint8_t *d_mask, const uint32_t *d_row_ids,
                                                    uint64_t rows, float *d_out);

/* lda_predict — LDA_impute (lda.cpp:421-590): argmax over classes of intercept + coef . [x,
 * onehot].  d_cat[n_cat] are the key columns in training order WITHOUT the label.  The reference
 * returns the class INDEX (lda.cpp:560); emit_label != 0 writes the class key instead, which is
 * what an in-place imputation needs.  d_mask as above. */
cofactor_status cofactor_lda_predict_device(cofactor_ctx *ctx, const float *params,
                                            uint64_t n_params, int32_t normalize,
                                            int32_t emit_label, const float *const *d_num,
                                            int32_t n_num, const int32_t *const *d_cat,
                                            int32_t n_cat, const uint8_t *d_mask, uint64_t rows,
                                            int32_t *d_out);

/* The same two over host columns (a DuckDB DataChunk): staged to the device, predicted there,
 * copied back; every row is written. */
cofactor_status cofactor_linreg_predict_host(cofactor_ctx *ctx, const float *params,
                                             uint64_t n_params, int32_t noise, int32_t normalize,
                                             uint64_t seed, const float *const *num,
                                             int32_t n_num, const int32_t *const *cat,
                                             int32_t n_cat, uint64_t rows, float *out);
cofactor_status cofactor_lda_predict_host(cofactor_ctx *ctx, const float *params,
                                          uint64_t n_params, int32_t normalize,
                                          int32_t emit_label, const float *const *num,
                                          int32_t n_num, const int32_t *const *cat, int32_t n_cat,
                                          uint64_t rows, int32_t *out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* COFACTOR_HIP_H */
