// gg_extension.hpp — shared between the table functions (gg_duckdb_extension.cpp) and the planner rules
// (gg_plan_rule.cpp) of the loadable extension.
#pragma once

#include <functional>

#include "duckdb/function/table_function.hpp"
#include "gg_operators.hpp"

namespace duckdb {

//! What a gg scan holds while it runs: the device graph, the GG source operator and its state.
struct GGOpened {
	shared_ptr<GGGraph> graph;
	unique_ptr<PhysicalOperator> source;
	unique_ptr<GlobalSourceState> gstate;
};

//! Bind data of every gg table function: a recipe that scans the base tables, builds the graph and
//! creates the source operator.  It runs when the scan is initialised (once per execution, never during
//! EXPLAIN or PREPARE), the point where the reference's hash-join build pipelines would run.
struct GGFunctionData : public TableFunctionData {
	std::function<void(ClientContext &, GGOpened &)> open;
	string description;           // what EXPLAIN prints under the operator name
	bool parallel_result = false; // the result is large: let several pipeline threads drain it
};

//! The table function whose init runs GGFunctionData::open and whose function is the source's GetData.
//! `name` is what EXPLAIN shows for the PhysicalTableScan wrapped around it.
TableFunction GGScanFunction(const string &name, vector<LogicalType> arguments = {},
                             table_function_bind_t bind = nullptr);

class TableCatalogEntry;

//! One base-table scan feeding a sink: either the table's storage read directly — row-group morsels on the
//! reference's worker threads, in the query's own transaction (gg_ingest.cpp) — or, for anything that is
//! not a plain table (views, arbitrary SQL), a statement run on a side connection.
struct GGScanSource {
	TableCatalogEntry *table = nullptr;
	vector<column_t> columns; // table column ids (COLUMN_IDENTIFIER_ROW_ID allowed)
	string sql;               // used when table is null
	bool Empty() const {
		return !table && sql.empty();
	}
};

//! The scans a graph is built from.
struct GGGraphSpec {
	GGScanSource vertices; // key column; empty: vertex set = distinct endpoint ids of the edges
	GGScanSource edges;    // (src, dst[, rowid])
	bool edges_with_rowid = false; // (SQL sources: the statement's third column is the edge's rowid)
	int shards = 1; // > 1: ownership-sharded over that many device contexts (GGGraph::peers)
};

//! Push every row of `source` through `sink` the way a pipeline would: GetGlobalSinkState, Sink per
//! <=1024-row chunk (concurrently for table sources), Combine, Finalize.
void GGRunSinkPipeline(ClientContext &context, const GGScanSource &source, PhysicalOperator &sink);

//! every non-NULL value of a one-column source, through the same ingest path (order unspecified)
vector<int64_t> GGScanInt64Column(ClientContext &context, const GGScanSource &source);

//! (schema.)table resolved in the catalog, or a SQL fallback `SELECT columns FROM name` for views
GGScanSource GGTableSource(ClientContext &context, const string &table_name, const vector<string> &columns,
                           bool with_rowid);
//! Builds the graph — or hands back the pinned one for exactly these tables and columns (gg_graph_pin), if this
//! connection asked for pinned graphs (PRAGMA gg_use_pinned_graphs) and its transaction has changed nothing.
shared_ptr<GGGraph> GGBuildGraph(ClientContext &context, const GGGraphSpec &spec);

//! Per-connection switches (the reference's pragmas act on one ClientContext, client_context.hpp:61-95; so do
//! ours).  Entries die with their connection: they are held by weak_ptr.
struct GGConnectionFlags {
	bool rules = false;        // PRAGMA enable_gpu_graph / disable_gpu_graph
	bool pinned_graphs = false; // PRAGMA gg_use_pinned_graphs / gg_ignore_pinned_graphs
	bool joins = false;         // PRAGMA enable_gpu_joins / disable_gpu_joins: ANY single-key inner join over a table scan
};
GGConnectionFlags GGGetConnectionFlags(ClientContext &context);
void GGSetConnectionFlags(ClientContext &context, const GGConnectionFlags &flags);
//! a statement is about to change `table_oid`: graphs pinned on it are dropped
void GGDropPinsOfTable(idx_t table_oid);

//! first column of `sql` (run on a side connection) as int64, NULLs skipped
vector<int64_t> GGQueryInt64Column(ClientContext &context, const string &sql, const char *what);

//! "identifier" with embedded quotes doubled
string GGQuote(const string &ident);

//! gg_plan_rule.cpp: hand the planner rules to the interposition shim if it is loaded; registers
//! `PRAGMA enable_gpu_graph` / `PRAGMA disable_gpu_graph`.
void GGRegisterPlanRules(ClientContext &context);

} // namespace duckdb
