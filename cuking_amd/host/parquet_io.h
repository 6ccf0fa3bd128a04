// Parquet boundary of the `cuking` binary, on Arrow/Parquet C++ (the reference
// links Arrow 8.0.0, Dockerfile:116; here the copy bundled with pyarrow).
//   input : cuking.cu:574-672 -- three columns BY POSITION, INT64 / INT64 /
//           INT32 (row_idx, col_idx, n_alt_alleles), all row groups.
//   output: cuking.cu:767-863 -- i, j (String), kin (FLOAT), ibs0/1/2 (INT32),
//           all REQUIRED, SNAPPY, one row group.
#ifndef CUKING_AMD_HOST_PARQUET_IO_H_
#define CUKING_AMD_HOST_PARQUET_IO_H_

#include <cstdint>
#include <string>
#include <vector>

#include "cuking_amd.h"

namespace cuking_host {

struct Triples {
  std::vector<int64_t> row_idx, col_idx;
  std::vector<int32_t> n_alt_alleles;
};

// Reads one input table -- all of it (row_group < 0), or one of its row groups,
// so that the unit of parallelism of the decode is the row group, not the file
// (the reference parallelises over files only, cuking.cu:550-553).  Spark writes
// OPTIONAL columns: those are accepted; a null row_idx / col_idx is an error, a
// null n_alt_alleles drops the entry (= missing genotype).  Returns "" or the
// error message.
std::string ReadTriples(const std::string &path, int row_group, Triples *out);
// The same table (or row group) in batches of at most `batch_rows` triples, handed to
// `sink` as they are decoded: the three column readers advance in step, the batch
// buffers (`scratch`, reused from call to call) stay in the cache, and no vector of a
// whole column chunk -- 20 bytes per triple, first zero-filled, then written, then
// read -- is ever allocated.  Same acceptance rules and messages as ReadTriples; null
// genotypes are dropped before the sink sees the batch.  `sink` returns "" or an
// error that ends the read.
struct TripleSink {
  virtual ~TripleSink() = default;
  virtual std::string Consume(const int64_t *row_idx, const int64_t *col_idx,
                              const int32_t *n_alt_alleles, size_t n) = 0;
};
struct TripleScratch {
  std::vector<int64_t> row_idx, col_idx;
  std::vector<int32_t> n_alt_alleles;
  std::vector<int16_t> def_levels;
};
std::string StreamTriples(const std::string &path, int row_group, size_t batch_rows,
                          TripleScratch *scratch, TripleSink *sink);
// Number of row groups of a table (reads the footer only; validates the schema
// like ReadTriples).
std::string CountRowGroups(const std::string &path, int *num_row_groups);

// Writes `results[0..n)` (already sorted) with sample ids looked up by index.
std::string WriteResults(const std::string &path, const cuking_result *results,
                         size_t n, const std::vector<std::string> &sample_ids,
                         uint64_t *bytes_written);

}  // namespace cuking_host

#endif  // CUKING_AMD_HOST_PARQUET_IO_H_
