/*
 * cuking_amd.h -- C ABI of the MI355X-native KING hot path.
 *
 * The reference (populationgenomics/cuKING, cuking.cu) has no library or FFI
 * boundary: its one kernel is launched inline from Run() (cuking.cu:734-741).
 * This header is the boundary a maintainer would bind instead.  Each entry
 * point cites the reference lines it replaces.  Conventions:
 *
 *   - extern "C", plain pointers and sizes, POD structs only; no C++ or torch
 *     types cross the boundary.
 *   - every function returns a cuking_status (0 = OK) unless it is a pure
 *     size/index helper; cuking_last_error() gives the message.
 *   - pointers named d_* are DEVICE pointers (hipMalloc'd, or e.g. a torch
 *     tensor's data_ptr()); everything else is host memory owned by the caller.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *     Calls enqueue work on it and return without synchronising unless stated.
 *   - one context per GPU; a context is used by one host thread at a time.
 *   - there is no CPU fallback: without a usable gfx950 device every device
 *     entry point fails with CUKING_ERR_DEVICE.
 */
#ifndef CUKING_AMD_H_
#define CUKING_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUKING_ABI_VERSION 2

typedef enum cuking_status {
  CUKING_OK = 0,
  CUKING_ERR_INVALID_ARGUMENT = 1,    /* absl::InvalidArgument, cuking.cu:437-462 */
  CUKING_ERR_FAILED_PRECONDITION = 2, /* e.g. bad n_alt_alleles, cuking.cu:698-702 */
  CUKING_ERR_RESOURCE_EXHAUSTED = 3,  /* result overflow, cuking.cu:747-751 */
  CUKING_ERR_OUT_OF_MEMORY = 4,       /* cuking.cu:113-118 */
  CUKING_ERR_DEVICE = 5               /* any HIP failure (unchecked in the reference, :738-744) */
} cuking_status;

/* cuking.cu:129-179 (struct Submatrix): one block of the upper-triangular
 * block matrix of sample pairs.  Samples are stored rows first, then columns;
 * a diagonal block stores its samples once. */
typedef struct cuking_submatrix {
  uint32_t i_begin, i_end; /* sample row range    */
  uint32_t j_begin, j_end; /* sample column range */
} cuking_submatrix;

/* cuking.cu:182-186 (struct KingResult), 24 bytes. */
typedef struct cuking_result {
  uint32_t sample_i, sample_j;
  float kin;
  uint32_t ibs0, ibs1, ibs2;
} cuking_result;

/* The six per-pair sums of cuking.cu:216-240 (diagnostic output only). */
typedef struct cuking_counts {
  uint32_t het_i, het_j, both_het, opposing_hom, concordant_hom, shared;
} cuking_counts;

/* ------------------------------------------------------------------------ */
/* Host-only helpers (no GPU touched).                                       */
/* ------------------------------------------------------------------------ */

/* cuking.cu:130-152 + flag validation :455-462.  Ranges are clamped to
 * num_samples (the reference wraps when block*size > N). */
cuking_status cuking_submatrix_init(cuking_submatrix *sm, uint32_t num_samples,
                                    uint32_t split_factor,
                                    uint32_t shard_index);
uint32_t cuking_submatrix_num_rows(const cuking_submatrix *sm);    /* :154 */
uint32_t cuking_submatrix_num_cols(const cuking_submatrix *sm);    /* :156 */
uint32_t cuking_submatrix_num_samples(const cuking_submatrix *sm); /* :159-162 */
uint32_t cuking_submatrix_contains(const cuking_submatrix *sm, uint32_t index);      /* :165-168 */
uint32_t cuking_submatrix_sample_offset(const cuking_submatrix *sm, uint32_t index); /* :171-175 */
/* Number of (i < j) pairs the block holds = what the kernel evaluates (:199). */
uint64_t cuking_submatrix_num_pairs(const cuking_submatrix *sm);

uint32_t cuking_padded_sites(uint32_t num_sites);     /* :498-500 (x32) */
uint32_t cuking_words_per_sample(uint32_t num_sites); /* :513 */
/* Algorithmic bytes one pair reads: 2 samples x words_per_sample x 8 (:209-224). */
uint64_t cuking_bytes_per_pair(uint32_t words_per_sample);

/* cuking.cu:675-703 + :317-323 on HOST memory: clears bits of an all-ones
 * (cuking.cu:523) bitset for each triple whose sample is in the block.
 * Relaxed atomic ANDs, so concurrent calls on one bitset from several reader
 * threads are safe (cuking.cu:550-553).  FAILED_PRECONDITION for n_alt outside
 * {0,1,2}; INVALID_ARGUMENT for a row_idx outside the padded sites. */
cuking_status cuking_pack_host(const cuking_submatrix *sm,
                               uint32_t words_per_sample, uint64_t *bit_set,
                               const int64_t *row_idx, const int64_t *col_idx,
                               const int32_t *n_alt_alleles,
                               size_t num_triples);

/* Host half of the compact device pack (below): filters triples to the block
 * (cuking.cu:677-679), validates them exactly like cuking_pack_host (same
 * status codes and messages) and writes, for each one kept, its site index and
 * its block-local sample offset (cuking.cu:171-175) with n_alt in bits 30..31:
 * 8 bytes per genotype for the trip to the GPU instead of the 20 of the three
 * Parquet columns.  site / sample_alt hold num_triples entries; *num_out =
 * entries written.  Thread-safe (no shared state). */
cuking_status cuking_narrow_triples(const cuking_submatrix *sm,
                                    uint32_t words_per_sample,
                                    const int64_t *row_idx, const int64_t *col_idx,
                                    const int32_t *n_alt_alleles, size_t num_triples,
                                    uint32_t *site, uint32_t *sample_alt,
                                    size_t *num_out);

/* Message of the calling thread's most recent failing call ("" if none). */
const char *cuking_last_error(void);
uint32_t cuking_abi_version(void);

/* ------------------------------------------------------------------------ */
/* Device context and memory.                                                */
/* ------------------------------------------------------------------------ */
typedef struct cuking_ctx cuking_ctx;

int cuking_device_count(void);
/* Binds a context to HIP device `device` (must be gfx950). */
cuking_status cuking_ctx_create(int device, cuking_ctx **out);
void cuking_ctx_destroy(cuking_ctx *ctx);

/* Explicit device memory for hosts without their own allocator (replaces
 * cudaMallocManaged, cuking.cu:109-120: no managed memory on the hot path). */
cuking_status cuking_device_alloc(cuking_ctx *ctx, size_t bytes, void **d_ptr);
cuking_status cuking_device_free(cuking_ctx *ctx, void *d_ptr);
cuking_status cuking_memset_async(cuking_ctx *ctx, void *d_ptr, int byte_value,
                                  size_t bytes, void *stream);
cuking_status cuking_copy_to_device(cuking_ctx *ctx, void *d_dst,
                                    const void *src, size_t bytes, void *stream);
cuking_status cuking_copy_to_host(cuking_ctx *ctx, void *dst, const void *d_src,
                                  size_t bytes, void *stream);
cuking_status cuking_stream_synchronize(cuking_ctx *ctx, void *stream);
/* Extra streams for hosts without their own (e.g. one per Parquet reader
 * thread).  The memory, copy, stream and cuking_pack_device entry points may
 * be called from several host threads at once, each on its own stream; the
 * compute / prepare / timing entry points need one caller at a time. */
cuking_status cuking_stream_create(cuking_ctx *ctx, void **stream);
cuking_status cuking_stream_destroy(cuking_ctx *ctx, void *stream);
/* Events, for hosts that pipeline several staging buffers through one stream
 * (wait for ONE earlier piece of work instead of the whole stream). */
cuking_status cuking_event_create(cuking_ctx *ctx, void **event);
cuking_status cuking_event_record(cuking_ctx *ctx, void *event, void *stream);
cuking_status cuking_event_synchronize(cuking_ctx *ctx, void *event);
cuking_status cuking_event_destroy(cuking_ctx *ctx, void *event);
/* Page-locked host memory for staging buffers. */
cuking_status cuking_host_alloc(cuking_ctx *ctx, size_t bytes, void **ptr);
cuking_status cuking_host_free(cuking_ctx *ctx, void *ptr);

/* ------------------------------------------------------------------------ */
/* The hot path.                                                             */
/* ------------------------------------------------------------------------ */

/* Pack on the device (cuking.cu:675-703 as a kernel): d_bit_set must already
 * be all ones (cuking.cu:523; cuking_memset_async(.., 0xFF, ..)).  Triples
 * live in device memory.  *d_status (one u32, zeroed by the caller) receives
 * a bit mask: 1 = n_alt outside {0,1,2}, 2 = row_idx out of range. */
cuking_status cuking_pack_device(cuking_ctx *ctx, const cuking_submatrix *sm,
                                 uint32_t words_per_sample, uint64_t *d_bit_set,
                                 const int64_t *d_row_idx,
                                 const int64_t *d_col_idx,
                                 const int32_t *d_n_alt_alleles,
                                 size_t num_triples, uint32_t *d_status,
                                 void *stream);

/* The same for triples prepared by cuking_narrow_triples (device copies of its
 * two output arrays).  *d_status as above (2 also flags a sample offset outside
 * the block). */
cuking_status cuking_pack_device_compact(cuking_ctx *ctx, const cuking_submatrix *sm,
                                         uint32_t words_per_sample,
                                         uint64_t *d_bit_set, const uint32_t *d_site,
                                         const uint32_t *d_sample_alt,
                                         size_t num_triples, uint32_t *d_status,
                                         void *stream);

/* Which device kernel evaluates the pairs. */
typedef enum cuking_kernel {
  CUKING_KERNEL_TILED = 0,  /* LDS-staged tile kernels: matrix-core variant (default) or VALU popcount variants */
  CUKING_KERNEL_STREAM = 1  /* one pair per wavefront, wave-level reductions */
} cuking_kernel;
cuking_status cuking_ctx_set_kernel(cuking_ctx *ctx, cuking_kernel kernel);
/* Tuning knobs of the tiled kernel: "variant" (compiled kernel shape, 0 ..
 * cuking_num_variants()-1; also env CUKING_AMD_VARIANT; 0..4 are VALU
 * AND/popcount shapes, 5, 6 and 7 the matrix-core kernels: 5 = five plane
 * products on the reference's two bit planes, 6 = four plane products on one
 * fp4 code per site, 7 = the default: ONE plane product per pair as a rigorous
 * upper bound on kinship, and the reference's exact sums for the few pairs
 * that bound lets through (one wavefront per candidate pair; quadrants with
 * many candidates go to kernel 6) -- same records for any data, the bound only
 * decides who computes a pair exactly.  It applies to the lean form with
 * 0 < kin_threshold < 1/2; otherwise, and for the diagnostic counts, variant 7
 * runs kernel 6 on the quadrants of its tiles.  7 has 256-sample tiles, all
 * others 128 or 64 (cuking_tile_samples).  6 and 7 serve bitsets below 2^22
 * sites and hand wider ones to 5, which hands bitsets from 2^24 sites on to
 * VALU shape 2; the tile edge stays the context variant's), "split_wgs"
 * (matrix-core variant: short launches cut their remainder of tiles into this
 * many equal pieces, default one per CU, 0 = never), "band_rows"
 * (tile-rows per scheduling band, 1..64, 0 = chosen by block size, the default;
 * env CUKING_AMD_BAND_ROWS), "xcd_swizzle" (matrix-core variant: the
 * workgroups resident on one XCD hold consecutive tiles of the band order --
 * 2 = patches of 32 tiles dealt round-robin to the XCDs (default), 1 = one
 * contiguous chunk per XCD, 0 = off; env CUKING_AMD_XCD_SWIZZLE),
 * "dyn_tail_tiles" (matrix-core variant: launches of at least this many tiles
 * hand their last ~6 % out through a counter instead of by workgroup index, so
 * that the XCDs, which differ by 2-3 %, finish together; default 16384, 0 =
 * never; env CUKING_AMD_DYN_TAIL_TILES) and
 * "counts_mode" (0 = lean: four sums per pair in the main loop, the hom/hom
 * count behind IBS2 recounted only for emitted pairs; 1 = full: all five sums
 * for every pair; -1 = automatic: lean when kin_threshold > c / sqrt(sites), c = 2.05 (1.6 for the VALU variants),
 * i.e. when few pairs are expected to pass).  Test hooks of variant 7:
 * "filter_quadrant_cap" (candidates per 128 x 128 quadrant beyond which the
 * quadrant goes to kernel 6, default 384), "filter_cand_cap" (entries of the
 * candidate list per launch chunk, default 2^25) and "filter_split_min_steps" (k-steps
 * of 256 sites a piece of a short launch's remainder must have, default 8).
 * Results do not depend on any of them. */
cuking_status cuking_ctx_set_option(cuking_ctx *ctx, const char *key,
                                    int64_t value);
/* Current value of "variant", "split_wgs", "band_rows", "xcd_swizzle",
 * "dyn_tail_tiles", "counts_mode", "reuse_prepared", "filter_quadrant_cap",
 * "filter_cand_cap" or "filter_split_min_steps"; diagnostics of variant 7 that WAIT for the device:
 * "filter_candidates" (pairs its bound has let through to the exact recount so
 * far) and "filter_dense_quadrants" (128 x 128 quadrants it has handed to kernel
 * 6 so far); read-only counters
 * "workspace_allocations", "host_syncs", "conversions_skipped". */
cuking_status cuking_ctx_get_option(const cuking_ctx *ctx, const char *key,
                                    int64_t *value);
int cuking_num_variants(void);
const char *cuking_variant_name(int variant);

/* ComputeKingKernel (cuking.cu:191-314) with its launch (:725-741): same
 * arguments, same meaning.  For every pair (i < j) of the block computes the
 * six masked popcount sums over d_bit_sets (layout cuking.cu:507-523: sample
 * s at d_bit_sets + SampleOffset(s) * words_per_sample, [het | hom_var]
 * planes), the float32 kinship (:289-294), and appends a cuking_result for
 * each pair with kin > kin_threshold (strict) at slot atomicAdd(d_result_index)
 * if that slot < max_results, else sets *d_result_overflow = 1 (:297-313).
 * d_result_index and d_result_overflow are NOT reset by the call (the caller
 * zeroes them, like cuking.cu:721-722), so several calls may append to one
 * buffer.  Record order is unspecified (sort afterwards, :761-765).
 * Asynchronous on `stream`.
 *
 * Numerics contract.  The sums, IBS0/1/2 and het counts are exact integers for
 * any width.  kin = fl32(0.5f + fl32(num / den)) with the IEEE-correct divide.
 * Below 2^22 sites (4,194,304; every BASELINE config is <= 200,000) every
 * partial sum of num = 2 bh - 4 opp - hi - hj is an integer below 2^24, so the
 * value is the same for every association order and every FMA contraction a
 * compiler may apply to cuking.cu:291-294: bit-exact against the reference.
 * (Variant 6 evaluates num as the integer hi + hj - 2 dd + 2 q, dd =
 * sites where both samples are defined, q = concordant - opposing homozygous
 * sites: the same integer, so the same float, below 2^22 sites; it is not used
 * beyond.  The default variant's records come from the reference's own six
 * sums and float expression, evaluated for every pair its bound admits; the
 * bound carries a margin for both float32 roundings (csrc/king_filter.hip).)
 * From 2^22 sites on, this library evaluates the expression left to
 * right with one float32 rounding per operation (no contraction); a reference
 * build that fuses multiply-adds may differ there in the last bit.  The
 * matrix-core variants count in float32 and serve bitsets up to 2^24 sites; wider
 * ones take a VALU variant automatically (same records).
 *
 * Streams.  The compute / prepare entry points convert the bitset into a
 * kernel-internal layout held by the context.  Calls on different streams of
 * one context are ordered by the library where a conversion would overwrite
 * what an earlier call's kernel may still read (event waits, no host
 * synchronisation); concurrent kernels only arise from
 * cuking_compute_king_rect launches on different streams. */
cuking_status cuking_compute_king(cuking_ctx *ctx, const cuking_submatrix *sm,
                                  uint32_t words_per_sample,
                                  const uint64_t *d_bit_sets,
                                  float kin_threshold, uint32_t max_results,
                                  cuking_result *d_results,
                                  uint32_t *d_result_index,
                                  uint32_t *d_result_overflow, void *stream);

/* Pair-space sharding inside one block (replaces multi-VM --split_factor
 * fan-out, cloud_batch_submit.py:45,73, for the GPUs of one node): the tiled
 * kernel enumerates the block's pairs as cuking_num_tiles() independent
 * square tiles; a rank evaluates tiles [tile_begin, tile_end).  The union of
 * disjoint ranges covering [0, num_tiles) equals cuking_compute_king(). */
uint64_t cuking_num_tiles(const cuking_ctx *ctx, const cuking_submatrix *sm);
uint32_t cuking_tile_samples(const cuking_ctx *ctx); /* samples per tile edge */
/* Which tile of the block tile index `tile` is: rows [*row_begin, *row_end)
 * x columns [*col_begin, *col_end) in global sample indices (clamped to the
 * block).  Host-only; lets a scheduler reason about the samples a tile range
 * touches.  The bounds are in LAYOUT order: the default kernel lays a block's samples
 * out sorted by their share of missing calls (option "filter_sort", default 1: whole-
 * block conversions; samples of equal share -- an ordinary cohort: all of them -- keep
 * their stored order), so in a cohort with low-call-rate samples the samples behind a
 * tile are not the ones these bounds name; the union over all tiles is the block either
 * way.  A caller that needs sample-accurate tiles sets "filter_sort" to 0. */
cuking_status cuking_tile_bounds(const cuking_ctx *ctx,
                                 const cuking_submatrix *sm, uint64_t tile,
                                 uint32_t *row_begin, uint32_t *row_end,
                                 uint32_t *col_begin, uint32_t *col_end);
/* (ctx may be NULL for these three: the default kernel shape is assumed.) */
cuking_status cuking_compute_king_tiles(
    cuking_ctx *ctx, const cuking_submatrix *sm, uint32_t words_per_sample,
    const uint64_t *d_bit_sets, uint64_t tile_begin, uint64_t tile_end,
    float kin_threshold, uint32_t max_results, cuking_result *d_results,
    uint32_t *d_result_index, uint32_t *d_result_overflow, void *stream);

/* The schedules of one block over the GPUs of a node (host arithmetic only; the
 * reference fans shards out over VMs instead: cloud_batch_submit.py:45,73).  ONE
 * implementation (cuking_amd/host/schedule.h) behind the C++ host `cuking --num_gpus=N`
 * and, through these entry points, the Python driver cuking_amd/dist.py.
 *   tile partition   contiguous ranges of the tile enumeration, equal or in proportion
 *                    to per-rank weights: out[2 r], out[2 r + 1] = rank r's [begin, end)
 *   chunk ranges     ascending tile-aligned sample chunks of the staged broadcast:
 *                    out[2 c], out[2 c + 1]; returns the number of chunks (<= num_chunks)
 *   staged steps     what rank `rank` does per chunk (tile rows dealt round-robin): six
 *                    words per chunk -- chunk begin, chunk end, has_rect, row begin, row
 *                    end, row step (samples); returns the number of chunks */
void cuking_schedule_tile_partition(uint64_t num_tiles, uint32_t world, uint64_t *out);
cuking_status cuking_schedule_weighted_tile_partition(uint64_t num_tiles, const double *weights,
                                                      uint32_t world, uint64_t *out);
uint64_t cuking_schedule_calibration_tiles(uint64_t num_tiles, uint32_t world);
uint32_t cuking_schedule_chunk_ranges(uint32_t num_samples, uint32_t tile, uint32_t num_chunks,
                                      uint32_t *out);
uint32_t cuking_schedule_staged_steps(uint32_t num_samples, uint32_t tile, uint32_t world,
                                      uint32_t rank, uint32_t num_chunks, uint32_t *out);

/* Staged form of the same operator for a DIAGONAL block (rows == columns),
 * used when the bitset arrives in pieces (e.g. a chunked RCCL broadcast):
 * cuking_prepare_samples() converts samples [sample_begin, sample_end) (global
 * indices, tile aligned except at the block end) into the context's kernel
 * layout; cuking_compute_king_rect() then evaluates the pairs (i < j) of rows
 * x columns [col_begin, col_end) whose samples have all been prepared, where
 * the rows are the tile rows starting at row_begin, row_begin + row_step, ...
 * below row_end (row_step = 0 or the tile edge: every row of the range; a
 * multiple of the tile edge: every n-th tile row, which is how the GPUs of a
 * node share the rows round-robin).  Rectangles that tile the upper triangle reproduce
 * cuking_compute_king() exactly.  Both are asynchronous on `stream`; kernels
 * of different rectangles may run concurrently on different streams (they only
 * read the prepared layout and append atomically). */
cuking_status cuking_prepare_samples(cuking_ctx *ctx, const cuking_submatrix *sm,
                                     uint32_t words_per_sample,
                                     const uint64_t *d_bit_sets,
                                     uint32_t sample_begin, uint32_t sample_end,
                                     void *stream);
cuking_status cuking_compute_king_rect(
    cuking_ctx *ctx, const cuking_submatrix *sm, uint32_t words_per_sample,
    const uint64_t *d_bit_sets, uint32_t row_begin, uint32_t row_end,
    uint32_t row_step, uint32_t col_begin, uint32_t col_end, float kin_threshold,
    uint32_t max_results, cuking_result *d_results, uint32_t *d_result_index,
    uint32_t *d_result_overflow, void *stream);

/* Sizes the context's workspace for `sm` up front: the kernel-internal layout
 * of the block (cuking.cu:513-523 sizes the reference's one buffer the same
 * way, before anything runs), the tile enumeration's prefix table and, for each
 * of the `num_streams` (<= 8) streams named, the remainder-split slab of the
 * matrix-core kernel.  Afterwards the compute / prepare calls for this block
 * (or a smaller one) on those streams allocate nothing and never wait for the
 * device -- which a host that drives several GPUs from one process needs once
 * collectives are in flight (host/multi_gpu.cc reserves before its first
 * broadcast).  May synchronise the device itself.  The context's
 * "workspace_allocations" / "host_syncs" options (cuking_ctx_get_option) count
 * the allocations and host-side waits made on behalf of the workspace so far. */
cuking_status cuking_ctx_reserve(cuking_ctx *ctx, const cuking_submatrix *sm,
                                 uint32_t words_per_sample, void *const *streams,
                                 size_t num_streams);

/* With the option "reuse_prepared" = 1 a compute / prepare call whose block,
 * width, kernel shape and bitset POINTER equal those of the layout the
 * workspace already holds launches the pair kernel only (the conversion is
 * 1.7 % of a 10k x 100k-site call).  The host thereby promises not to rewrite
 * that bitset in place without calling cuking_invalidate() before the next
 * compute call.  Default 0: like ComputeKingKernel (cuking.cu:191-195), every
 * call reads whatever the bitset holds when it runs. */
cuking_status cuking_invalidate(cuking_ctx *ctx);

/* Diagnostic: the six sums of every pair, no threshold.  d_counts holds
 * NumRows x NumCols records, pair (i, j) at [(i - i_begin) * NumCols +
 * (j - j_begin)]; entries with i >= j are left untouched. */
cuking_status cuking_compute_counts(cuking_ctx *ctx, const cuking_submatrix *sm,
                                    uint32_t words_per_sample,
                                    const uint64_t *d_bit_sets,
                                    cuking_counts *d_counts, void *stream);

/* cuking.cu:761-765 on host memory: sort by (sample_i, sample_j, kin). */
void cuking_sort_results(cuking_result *results, size_t num_results);

/* ------------------------------------------------------------------------ */
/* Measurement hooks (replace the StopWatch prints, cuking.cu:325-337).      */
/* ------------------------------------------------------------------------ */

/* When enabled, every launch of the pair kernel is bracketed by HIP events on
 * the stream it is launched on. */
cuking_status cuking_timing_enable(cuking_ctx *ctx, int enabled);
cuking_status cuking_timing_reset(cuking_ctx *ctx);
/* Synchronises the recorded events; returns total device milliseconds and
 * the number of launches since the last reset, for the pair kernel and for
 * the layout-preparation kernel that precedes it. */
cuking_status cuking_timing_collect(cuking_ctx *ctx, double *king_ms,
                                    uint64_t *king_launches, double *prepare_ms,
                                    uint64_t *prepare_launches);

/* Sustained shader clock while other work runs: enqueues ONE wavefront on
 * `stream` (use a stream of its own) that watches the shader-clock counter and
 * the constant 100 MHz counter for `microseconds` of wall time and then writes
 * d_ticks[0] = shader ticks, d_ticks[1] = 100 MHz ticks; clock in MHz =
 * 100 * d_ticks[0] / d_ticks[1].  It holds one wave slot of one CU meanwhile,
 * so it belongs in a pass of its own, not in a timed region. */
cuking_status cuking_clock_probe(cuking_ctx *ctx, uint64_t microseconds,
                                 uint64_t *d_ticks, void *stream);

/* ------------------------------------------------------------------------ */
/* Synthetic inputs for benchmarks (no reference counterpart).    */
/* ------------------------------------------------------------------------ */

/* Fills rows [sample_begin, sample_end) of a reference-layout bitset with
 * Hardy-Weinberg genotypes (per-site AF ~ U(0.05,0.5), 1 % missing) and the
 * planted relatives described by kind/pa/pb (device arrays of num_samples
 * u32 each; 0 founder, 1 duplicate of pa, 2 child of pa x pb).  Row 0 of
 * d_bit_set is sample_begin.  Bit-identical to oracle/synth_oracle.c. */
cuking_status cuking_synth_bitset(cuking_ctx *ctx, uint64_t seed,
                                  const uint32_t *d_kind, const uint32_t *d_pa,
                                  const uint32_t *d_pb, uint32_t sample_begin,
                                  uint32_t sample_end, uint32_t num_sites,
                                  uint32_t words_per_sample,
                                  uint64_t *d_bit_set, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CUKING_AMD_H_ */
