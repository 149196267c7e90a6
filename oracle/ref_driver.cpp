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
