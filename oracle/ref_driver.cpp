// ref_driver.cpp — bulk loader for the compiled reference (TEST INFRASTRUCTURE ONLY).
//
// Built only where /root/reference exists (oracle/Makefile target `ref`), against the reference's
// own headers, into oracle/_ref/libggref.so next to oracle/_ref/libduckdb.so.  It adds nothing to
// the reference's behaviour: it appends int64 columns to an existing table through the reference's
// public Appender (src/include/duckdb/main/appender.hpp:24-71), because feeding tens of millions of
// rows through the per-value C API from Python would dominate the baseline's set-up time.
// Queries themselves go through the reference's C API (duckdb_query, src/include/duckdb.h:286).
#include <cstdint>
#include <string>

#include "duckdb.hpp"
#include "duckdb/main/appender.hpp"

extern "C" int ggref_append_int64_columns(void *c_api_connection, const char *table, int ncols,
                                          const int64_t *const *cols, uint64_t nrows, char *err, int errlen) {
  try {
    // duckdb_connection is a duckdb::Connection* (src/main/capi/duckdb-c.cpp:42-60)
    auto *con = (duckdb::Connection *)c_api_connection;
    duckdb::Appender app(*con, table);
    for (uint64_t r = 0; r < nrows; r++) {
      app.BeginRow();
      for (int c = 0; c < ncols; c++) app.Append<int64_t>(cols[c][r]);
      app.EndRow();
    }
    app.Close();
    return 0;
  } catch (std::exception &e) {
    if (err && errlen > 0) {
      std::string m = e.what();
      size_t n = m.size() < (size_t)errlen - 1 ? m.size() : (size_t)errlen - 1;
      m.copy(err, n);
      err[n] = 0;
    }
    return 1;
  }
}

// A query's result as text, whatever its column types (the C API of this reference cannot return DECIMAL or HUGEINT
// columns): values by Value::ToString(), NULL as an empty field flagged in `nulls`, booleans as 1 / 0 — the
// conversion of the reference's own sqllogictest runner (test/sqlite/test_sqllogictest.cpp:306-336), for
// tests/test_reference_vectors.py.  Layout of *out (malloc'd, the caller frees it with ggref_free): fields separated
// by '\x1f', rows by '\x1e'; a NULL field is the single byte '\x00'.  Returns 0, or 1 with the error text in err.
#include <cstdlib>
#include <cstring>

extern "C" int ggref_query_text(void *c_api_connection, const char *sql, char **out, uint64_t *out_len, uint64_t *n_rows,
                                uint64_t *n_cols, char *err, int errlen) {
  auto fail = [&](const std::string &m) {
    if (err && errlen > 0) {
      size_t n = m.size() < (size_t)errlen - 1 ? m.size() : (size_t)errlen - 1;
      m.copy(err, n);
      err[n] = 0;
    }
    return 1;
  };
  try {
    auto *con = (duckdb::Connection *)c_api_connection;
    auto result = con->Query(sql);
    if (!result->success) return fail(result->error);
    std::string buf;
    const duckdb::idx_t rows = result->collection.Count(), cols = result->ColumnCount();
    for (duckdb::idx_t r = 0; r < rows; r++) {
      for (duckdb::idx_t c = 0; c < cols; c++) {
        auto value = result->GetValue(c, r);
        if (value.is_null) {
          buf.push_back('\0');
        } else if (result->types[c].id() == duckdb::LogicalTypeId::BOOLEAN) {
          buf += value.value_.boolean ? "1" : "0";
        } else {
          buf += value.ToString();
        }
        buf.push_back(c + 1 < cols ? '\x1f' : '\x1e');
      }
    }
    *out = (char *)malloc(buf.size() + 1);
    memcpy(*out, buf.data(), buf.size());
    (*out)[buf.size()] = 0;
    *out_len = buf.size();
    *n_rows = rows;
    *n_cols = cols;
    return 0;
  } catch (std::exception &e) {
    return fail(e.what());
  }
}

extern "C" void ggref_free(void *p) { free(p); }
