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
	string description; // what EXPLAIN prints under the operator name
};

//! The table function whose init runs GGFunctionData::open and whose function is the source's GetData.
//! `name` is what EXPLAIN shows for the PhysicalTableScan wrapped around it.
TableFunction GGScanFunction(const string &name, vector<LogicalType> arguments = {},
                             table_function_bind_t bind = nullptr);

//! The base-table scans a graph is built from (SQL run on a side connection, one chunk per Sink call).
struct GGGraphSpec {
	string vertex_sql; // SELECT key FROM vertex table; empty: vertex set = distinct endpoint ids of the edges
	string edge_sql;   // SELECT src, dst[, rowid] FROM edge table
};
shared_ptr<GGGraph> GGBuildGraph(ClientContext &context, const GGGraphSpec &spec);

//! first column of `sql` (run on a side connection) as int64, NULLs skipped
vector<int64_t> GGQueryInt64Column(ClientContext &context, const string &sql, const char *what);

//! "identifier" with embedded quotes doubled
string GGQuote(const string &ident);

//! gg_plan_rule.cpp: hand the planner rules to the interposition shim if it is loaded; registers
//! `PRAGMA enable_gpu_graph` / `PRAGMA disable_gpu_graph`.
void GGRegisterPlanRules(ClientContext &context);

} // namespace duckdb
