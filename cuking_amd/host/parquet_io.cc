#include "parquet_io.h"

#include <arrow/io/file.h>
#include <parquet/api/reader.h>
#include <parquet/api/writer.h>

#include <algorithm>
#include <memory>
#include <sstream>

namespace cuking_host {

namespace {

// Reads a whole column chunk into `values` (appending).  `keep` (optional)
// receives one flag per ROW: false where the value is null.
template <typename ReaderT, typename T>
std::string ReadColumn(parquet::ColumnReader *column, int64_t rows,
                       std::vector<T> *values, std::vector<uint8_t> *valid) {
  auto *reader = static_cast<ReaderT *>(column);
  const bool nullable = column->descr()->max_definition_level() > 0;
  if (column->descr()->max_repetition_level() > 0)
    return "repeated columns are not supported";
  const size_t base = values->size();
  values->resize(base + (size_t)rows);
  std::vector<int16_t> def;
  if (nullable) def.resize((size_t)rows);
  int64_t levels_done = 0, values_done = 0;
  while (reader->HasNext() && levels_done < rows) {
    int64_t values_read = 0;
    const int64_t levels_read = reader->ReadBatch(
        rows - levels_done, nullable ? def.data() + levels_done : nullptr,
        nullptr, values->data() + base + values_done, &values_read);
    if (levels_read == 0 && values_read == 0) break;
    levels_done += nullable ? levels_read : values_read;
    values_done += values_read;
  }
  if (levels_done != rows) return "column chunk shorter than its row group";
  if (!nullable) {
    if (valid) valid->insert(valid->end(), (size_t)rows, 1);
    return "";
  }
  if (values_done == rows) {  // OPTIONAL without nulls: the common Spark case
    if (valid) valid->insert(valid->end(), (size_t)rows, 1);
    return "";
  }
  if (valid == nullptr) return "null values are not allowed in this column";
  // Spread the compact values back over their rows (from the end).
  const int16_t max_def = column->descr()->max_definition_level();
  int64_t v = values_done;
  for (int64_t r = rows - 1; r >= 0; --r) {
    const bool ok = def[(size_t)r] == max_def;
    (*values)[base + (size_t)r] = ok ? (*values)[base + (size_t)(--v)] : T();
  }
  for (int64_t r = 0; r < rows; ++r)
    valid->push_back(def[(size_t)r] == max_def ? 1 : 0);
  return "";
}

}  // namespace

namespace {

// cuking.cu:585-590, :608-613, :630-635, :652-657: three columns by position,
// INT64 / INT64 / INT32.
std::string CheckSchema(const parquet::FileMetaData &meta, const std::string &path) {
  constexpr int kNumColumns = 3;
  if (meta.num_columns() != kNumColumns) {
    std::ostringstream os;
    os << "Expected " << kNumColumns << " columns, found " << meta.num_columns() << " in "
       << path;
    return os.str();
  }
  const parquet::Type::type want[kNumColumns] = {parquet::Type::INT64, parquet::Type::INT64,
                                                 parquet::Type::INT32};
  for (int c = 0; c < kNumColumns; ++c) {
    const auto got = meta.schema()->Column(c)->physical_type();
    if (got != want[c]) {
      std::ostringstream os;
      os << "Expected " << parquet::TypeToString(want[c]) << " type, found "
         << parquet::TypeToString(got) << " in " << path;
      return os.str();
    }
  }
  return "";
}

}  // namespace

std::string CountRowGroups(const std::string &path, int *num_row_groups) {
  try {
    std::unique_ptr<parquet::ParquetFileReader> file =
        parquet::ParquetFileReader::OpenFile(path, /*memory_map=*/false);
    const auto meta = file->metadata();
    const std::string err = CheckSchema(*meta, path);
    if (!err.empty()) return err;
    *num_row_groups = meta->num_row_groups();
    return "";
  } catch (const std::exception &e) {  // cuking.cu:580-583
    return std::string("Error reading ") + path + ": " + e.what();
  }
}

std::string ReadTriples(const std::string &path, int row_group, Triples *out) {
  try {
    std::unique_ptr<parquet::ParquetFileReader> file =
        parquet::ParquetFileReader::OpenFile(path, /*memory_map=*/false);
    const auto meta = file->metadata();
    {
      const std::string err = CheckSchema(*meta, path);
      if (!err.empty()) return err;
    }
    if (row_group >= meta->num_row_groups()) {
      std::ostringstream os;
      os << "row group " << row_group << " outside the " << meta->num_row_groups() << " of "
         << path;
      return os.str();
    }
    const int g_begin = row_group < 0 ? 0 : row_group;
    const int g_end = row_group < 0 ? meta->num_row_groups() : row_group + 1;
    int64_t expect = 0;
    for (int g = g_begin; g < g_end; ++g) expect += meta->RowGroup(g)->num_rows();
    out->row_idx.clear();
    out->col_idx.clear();
    out->n_alt_alleles.clear();
    out->row_idx.reserve((size_t)expect);
    out->col_idx.reserve((size_t)expect);
    out->n_alt_alleles.reserve((size_t)expect);
    std::vector<uint8_t> alt_valid;
    for (int g = g_begin; g < g_end; ++g) {
      auto group = file->RowGroup(g);
      const int64_t rows = group->metadata()->num_rows();
      std::string err;
      auto c0 = group->Column(0);
      err = ReadColumn<parquet::Int64Reader>(c0.get(), rows, &out->row_idx, nullptr);
      if (!err.empty()) return "row_idx: " + err + " in " + path;
      auto c1 = group->Column(1);
      err = ReadColumn<parquet::Int64Reader>(c1.get(), rows, &out->col_idx, nullptr);
      if (!err.empty()) return "col_idx: " + err + " in " + path;
      auto c2 = group->Column(2);
      err = ReadColumn<parquet::Int32Reader>(c2.get(), rows, &out->n_alt_alleles,
                                             &alt_valid);
      if (!err.empty()) return "n_alt_alleles: " + err + " in " + path;
    }
    // Drop entries whose genotype is null (they stay "missing").
    size_t w = 0;
    for (size_t r = 0; r < alt_valid.size(); ++r) {
      if (!alt_valid[r]) continue;
      out->row_idx[w] = out->row_idx[r];
      out->col_idx[w] = out->col_idx[r];
      out->n_alt_alleles[w] = out->n_alt_alleles[r];
      ++w;
    }
    out->row_idx.resize(w);
    out->col_idx.resize(w);
    out->n_alt_alleles.resize(w);
    return "";
  } catch (const std::exception &e) {  // cuking.cu:580-583
    return std::string("Error reading ") + path + ": " + e.what();
  }
}

namespace {

// Up to `want` rows of one column into `values` (compact: nulls leave no gap);
// `def` (nullable columns) receives one definition level per row.  Returns the
// rows read; `*values_read` the non-null values among them.
template <typename ReaderT, typename T>
int64_t ReadRows(ReaderT *reader, bool nullable, int64_t want, T *values, int16_t *def,
                 int64_t *values_read) {
  int64_t rows = 0, vals = 0;
  while (rows < want && reader->HasNext()) {
    int64_t got_values = 0;
    const int64_t got_levels = reader->ReadBatch(want - rows, nullable ? def + rows : nullptr,
                                                 nullptr, values + vals, &got_values);
    if (got_levels == 0 && got_values == 0) break;
    rows += nullable ? got_levels : got_values;
    vals += got_values;
  }
  *values_read = vals;
  return rows;
}

}  // namespace

std::string StreamTriples(const std::string &path, int row_group, size_t batch_rows,
                          TripleScratch *scratch, TripleSink *sink) {
  try {
    std::unique_ptr<parquet::ParquetFileReader> file =
        parquet::ParquetFileReader::OpenFile(path, /*memory_map=*/false);
    const auto meta = file->metadata();
    {
      const std::string err = CheckSchema(*meta, path);
      if (!err.empty()) return err;
    }
    if (row_group >= meta->num_row_groups()) {
      std::ostringstream os;
      os << "row group " << row_group << " outside the " << meta->num_row_groups() << " of "
         << path;
      return os.str();
    }
    if (batch_rows == 0) batch_rows = 1;
    scratch->row_idx.resize(batch_rows);
    scratch->col_idx.resize(batch_rows);
    scratch->n_alt_alleles.resize(batch_rows);
    scratch->def_levels.resize(batch_rows);
    int64_t *const row = scratch->row_idx.data();
    int64_t *const col = scratch->col_idx.data();
    int32_t *const alt = scratch->n_alt_alleles.data();
    int16_t *const def = scratch->def_levels.data();
    const int g_begin = row_group < 0 ? 0 : row_group;
    const int g_end = row_group < 0 ? meta->num_row_groups() : row_group + 1;
    for (int g = g_begin; g < g_end; ++g) {
      auto group = file->RowGroup(g);
      const int64_t rows = group->metadata()->num_rows();
      auto c0 = group->Column(0), c1 = group->Column(1), c2 = group->Column(2);
      const char *names[3] = {"row_idx", "col_idx", "n_alt_alleles"};
      parquet::ColumnReader *cols[3] = {c0.get(), c1.get(), c2.get()};
      bool nullable[3];
      for (int c = 0; c < 3; ++c) {
        if (cols[c]->descr()->max_repetition_level() > 0)
          return std::string(names[c]) + ": repeated columns are not supported in " + path;
        nullable[c] = cols[c]->descr()->max_definition_level() > 0;
      }
      auto *r0 = static_cast<parquet::Int64Reader *>(cols[0]);
      auto *r1 = static_cast<parquet::Int64Reader *>(cols[1]);
      auto *r2 = static_cast<parquet::Int32Reader *>(cols[2]);
      const int16_t alt_max_def = cols[2]->descr()->max_definition_level();
      for (int64_t done = 0; done < rows;) {
        const int64_t want = std::min<int64_t>((int64_t)batch_rows, rows - done);
        int64_t v0 = 0, v1 = 0, v2 = 0;
        if (ReadRows(r0, nullable[0], want, row, def, &v0) != want)
          return "row_idx: column chunk shorter than its row group in " + path;
        if (v0 != want) return "row_idx: null values are not allowed in this column in " + path;
        if (ReadRows(r1, nullable[1], want, col, def, &v1) != want)
          return "col_idx: column chunk shorter than its row group in " + path;
        if (v1 != want) return "col_idx: null values are not allowed in this column in " + path;
        if (ReadRows(r2, nullable[2], want, alt, def, &v2) != want)
          return "n_alt_alleles: column chunk shorter than its row group in " + path;
        size_t n = (size_t)want;
        if (v2 != want) {
          // Null genotypes (= missing): their rows leave the batch.  `alt` is compact
          // already (v2 values), row / col are closed up over the same rows.
          size_t w = 0;
          for (int64_t r = 0; r < want; ++r) {
            if (def[r] != alt_max_def) continue;
            row[w] = row[r];
            col[w] = col[r];
            ++w;
          }
          if ((int64_t)w != v2) return "n_alt_alleles: definition levels and values disagree in " + path;
          n = w;
        }
        if (n != 0) {
          const std::string err = sink->Consume(row, col, alt, n);
          if (!err.empty()) return err;
        }
        done += want;
      }
    }
    return "";
  } catch (const std::exception &e) {  // cuking.cu:580-583
    return std::string("Error reading ") + path + ": " + e.what();
  }
}

std::string WriteResults(const std::string &path, const cuking_result *results,
                         size_t n, const std::vector<std::string> &sample_ids,
                         uint64_t *bytes_written) {
  using parquet::Repetition;
  using parquet::schema::GroupNode;
  using parquet::schema::PrimitiveNode;
  try {
    parquet::schema::NodeVector fields;  // cuking.cu:770-788
    fields.push_back(PrimitiveNode::Make("i", Repetition::REQUIRED,
                                         parquet::LogicalType::String(),
                                         parquet::Type::BYTE_ARRAY));
    fields.push_back(PrimitiveNode::Make("j", Repetition::REQUIRED,
                                         parquet::LogicalType::String(),
                                         parquet::Type::BYTE_ARRAY));
    fields.push_back(PrimitiveNode::Make("kin", Repetition::REQUIRED,
                                         parquet::LogicalType::None(),
                                         parquet::Type::FLOAT));
    for (const char *name : {"ibs0", "ibs1", "ibs2"})
      fields.push_back(PrimitiveNode::Make(name, Repetition::REQUIRED,
                                           parquet::LogicalType::None(),
                                           parquet::Type::INT32));
    auto schema = std::static_pointer_cast<GroupNode>(
        GroupNode::Make("schema", Repetition::REQUIRED, fields));

    auto sink_result = arrow::io::FileOutputStream::Open(path);
    if (!sink_result.ok()) return sink_result.status().ToString();
    std::shared_ptr<arrow::io::FileOutputStream> sink = *sink_result;

    parquet::WriterProperties::Builder props;
    // "Hail's libhadoop doesn't support ZSTD" (cuking.cu:797-798).
    props.compression(parquet::Compression::SNAPPY);
    // One row group, like the reference (:804-805).
    props.max_row_group_length(std::max<int64_t>((int64_t)n, 1));
    auto writer = parquet::ParquetFileWriter::Open(sink, schema, props.build());
    parquet::RowGroupWriter *group = writer->AppendRowGroup();

    for (int side = 0; side < 2; ++side) {  // i, j (:807-826)
      auto *col = static_cast<parquet::ByteArrayWriter *>(group->NextColumn());
      std::vector<parquet::ByteArray> vals(n);
      for (size_t r = 0; r < n; ++r) {
        const uint32_t idx = side == 0 ? results[r].sample_i : results[r].sample_j;
        if (idx >= sample_ids.size()) return "result sample index out of range";
        const std::string &s = sample_ids[idx];
        vals[r] = parquet::ByteArray((uint32_t)s.size(),
                                     reinterpret_cast<const uint8_t *>(s.data()));
      }
      if (n) col->WriteBatch((int64_t)n, nullptr, nullptr, vals.data());
    }
    {  // kin (:827-833)
      auto *col = static_cast<parquet::FloatWriter *>(group->NextColumn());
      std::vector<float> vals(n);
      for (size_t r = 0; r < n; ++r) vals[r] = results[r].kin;
      if (n) col->WriteBatch((int64_t)n, nullptr, nullptr, vals.data());
    }
    for (int k = 0; k < 3; ++k) {  // ibs0, ibs1, ibs2 (:834-860)
      auto *col = static_cast<parquet::Int32Writer *>(group->NextColumn());
      std::vector<int32_t> vals(n);
      for (size_t r = 0; r < n; ++r) {
        const uint32_t v = k == 0 ? results[r].ibs0
                           : k == 1 ? results[r].ibs1 : results[r].ibs2;
        vals[r] = (int32_t)v;
      }
      if (n) col->WriteBatch((int64_t)n, nullptr, nullptr, vals.data());
    }
    writer->Close();
    auto pos = sink->Tell();
    if (bytes_written) *bytes_written = pos.ok() ? (uint64_t)*pos : 0;
    auto st = sink->Close();
    if (!st.ok()) return st.ToString();
    return "";
  } catch (const std::exception &e) {
    return std::string("Error writing ") + path + ": " + e.what();
  }
}

}  // namespace cuking_host
