// gg_duckdb_extension.cpp — zero-patch SQL surface for the GPU graph operators.
//
// Loaded with   LOAD '<repo>/duckdb_pgq_amd/gg_duckdb.duckdb_extension';
// (PhysicalLoad: dlopen + <basename>_init / <basename>_version,
//  src/execution/operator/helper/physical_load.cpp:29-70 of the reference).  It registers four table
// functions that the planner wraps in an ordinary PhysicalTableScan, so no reference
// file changes:
//
//   gg_khop(vertex_table, vertex_key, edge_table, src_col, dst_col, k_min, k_max)
//        -> (hops INTEGER, v0 BIGINT, ..., v{k_max} BIGINT)      all walks, NULL-padded
//   gg_khop_count(vertex_table, vertex_key, edge_table, src_col, dst_col, k_min, k_max)
//        -> (hops INTEGER, rows BIGINT, digest BIGINT, traversed_edges BIGINT)
//   gg_shortest_path(vertex_table, vertex_key, edge_table, src_col, dst_col, sources_sql, max_hops)
//        -> (startPerson BIGINT, friend BIGINT, hopCount INTEGER)
//   gg_same_neighbour_paths(vertices_sql, sources_sql, path_table, path_src, path_dst,
//                           filter_table, filter_src, filter_dst, hops)
//        -> (w BIGINT, v0 BIGINT, ..., v{hops} BIGINT)      Train Benchmark ConnectedSegments
//
// Each function runs the operator classes of gg_operators.hpp exactly the way the reference's
// PipelineExecutor would (pipeline_executor.cpp:47-131): source chunks -> Sink (per <=1024-row
// DataChunk) -> Combine -> Finalize, then GetData until an empty chunk.  Binding only fixes the schema;
// the base-table scans and the device work happen when the scan is initialised, i.e. at execution time.
// The same machinery serves the planner rules of gg_plan_rule.cpp, which put these scans in place of
// hash-join chains over an edge table (plan-level substitution).
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "duckdb.hpp"
#include "duckdb/catalog/catalog.hpp"
#include "duckdb/common/exception.hpp"
#include "duckdb/function/table_function.hpp"
#include "duckdb/main/client_context.hpp"
#include "duckdb/main/connection.hpp"
#include "duckdb/parallel/parallel_state.hpp"
#include "duckdb/parallel/thread_context.hpp"
#include "duckdb/parser/parsed_data/create_table_function_info.hpp"
#include "gg_extension.hpp"

namespace duckdb {

string GGQuote(const string &ident) {
	string out = "\"";
	for (auto c : ident) {
		if (c == '"') {
			out += "\"\"";
		} else {
			out += c;
		}
	}
	return out + "\"";
}

//! GG_TIMING=1 in the environment: phase times of every scan on stderr
static bool TimingEnabled() {
	static const bool enabled = std::getenv("GG_TIMING") != nullptr;
	return enabled;
}

struct PhaseTimer {
	std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
	void Lap(const char *what) {
		if (!TimingEnabled()) {
			return;
		}
		auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[gg] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
		last = now;
	}
};

shared_ptr<GGGraph> GGBuildGraph(ClientContext &context, const GGGraphSpec &spec) {
	PhaseTimer timer;
	auto graph = make_shared<GGGraph>(0);
	timer.Lap("device context");
	const bool derive = spec.vertices.Empty();
	if (!derive) {
		PhysicalGGVertexSink vsink(graph, {LogicalType::BIGINT}, 0);
		GGRunSinkPipeline(context, spec.vertices, vsink);
		timer.Lap("vertex table ingest");
	}
	PhysicalGGEdgeSink esink(graph, {LogicalType::BIGINT, LogicalType::BIGINT}, 0, false, derive);
	GGRunSinkPipeline(context, spec.edges, esink);
	timer.Lap("edge ingest + CSR build");
	return graph;
}

//! (vertex_table, vertex_key, edge_table, src_col, dst_col) arguments -> scans, resolved at execution time
struct GraphArguments {
	string vertex_table, vertex_key, edge_table, edge_src, edge_dst;
	explicit GraphArguments(vector<Value> &inputs)
	    : vertex_table(inputs[0].ToString()), vertex_key(inputs[1].ToString()), edge_table(inputs[2].ToString()),
	      edge_src(inputs[3].ToString()), edge_dst(inputs[4].ToString()) {
	}
	GGGraphSpec Resolve(ClientContext &context) const {
		GGGraphSpec spec;
		spec.vertices = GGTableSource(context, vertex_table, {vertex_key}, false);
		spec.edges = GGTableSource(context, edge_table, {edge_src, edge_dst}, false);
		return spec;
	}
};

static GGScanSource Statement(const string &sql) {
	GGScanSource source;
	source.sql = sql;
	return source;
}

//! Per-thread scan state: the thread's window onto the result (LocalSourceState of the source operator).
struct GGOperatorData : public FunctionOperatorData {
	explicit GGOperatorData(ClientContext &context) : thread(context), execution(context, thread) {
	}
	GGOpened *opened = nullptr;
	GGOpened own; // sequential scans open the graph themselves; parallel ones share GGParallelState's
	ThreadContext thread;
	ExecutionContext execution;
	unique_ptr<LocalSourceState> local;
};

//! Shared by the threads of a parallel scan: the opened graph and the source's global state.
struct GGParallelState : public ParallelState {
	GGOpened opened;
};

static void Open(ClientContext &context, const FunctionData *bind_data, GGOpened &opened) {
	auto &data = (GGFunctionData &)*bind_data;
	PhaseTimer timer;
	data.open(context, opened);
	opened.gstate = opened.source->GetGlobalSourceState(context);
	timer.Lap("scan opened (total)");
}

static unique_ptr<FunctionOperatorData> GGInit(ClientContext &context, const FunctionData *bind_data,
                                               const vector<column_t> &column_ids, TableFilterCollection *filters) {
	auto state = make_unique<GGOperatorData>(context);
	Open(context, bind_data, state->own);
	state->opened = &state->own;
	state->local = state->opened->source->GetLocalSourceState(state->execution, *state->opened->gstate);
	return move(state);
}

static void GGFunction(ClientContext &context, const FunctionData *bind_data_p, FunctionOperatorData *operator_state,
                       DataChunk *input, DataChunk &output) {
	auto &state = (GGOperatorData &)*operator_state;
	state.opened->source->GetData(state.execution, output, *state.opened->gstate, *state.local);
}

// ---- parallel scan: the reference's pipeline tasks drain one device-resident result together, each
// through its own pinned slab (PhysicalTableScan's parallel protocol, physical_table_scan.cpp:22-110)
static idx_t GGMaxThreads(ClientContext &context, const FunctionData *bind_data) {
	return ((const GGFunctionData &)*bind_data).parallel_result ? 8 : 1;
}

static unique_ptr<ParallelState> GGInitParallelState(ClientContext &context, const FunctionData *bind_data,
                                                     const vector<column_t> &column_ids,
                                                     TableFilterCollection *filters) {
	auto state = make_unique<GGParallelState>();
	Open(context, bind_data, state->opened);
	return move(state);
}

static unique_ptr<FunctionOperatorData> GGParallelInit(ClientContext &context, const FunctionData *bind_data,
                                                       ParallelState *parallel_state,
                                                       const vector<column_t> &column_ids,
                                                       TableFilterCollection *filters) {
	auto state = make_unique<GGOperatorData>(context);
	state->opened = &((GGParallelState &)*parallel_state).opened;
	state->local = state->opened->source->GetLocalSourceState(state->execution, *state->opened->gstate);
	return move(state);
}

static void GGParallelFunction(ClientContext &context, const FunctionData *bind_data,
                               FunctionOperatorData *operator_state, DataChunk *input, DataChunk &output,
                               ParallelState *parallel_state) {
	GGFunction(context, bind_data, operator_state, input, output);
}

static bool GGParallelStateNext(ClientContext &context, const FunctionData *bind_data, FunctionOperatorData *state,
                                ParallelState *parallel_state) {
	return false; // GetData claims its own slabs; an empty chunk means the result is exhausted
}

static string GGToString(const FunctionData *bind_data) {
	return ((const GGFunctionData &)*bind_data).description;
}

TableFunction GGScanFunction(const string &name, vector<LogicalType> arguments, table_function_bind_t bind) {
	TableFunction function(name, move(arguments), GGFunction, bind, GGInit);
	function.to_string = GGToString;
	function.max_threads = GGMaxThreads;
	function.init_parallel_state = GGInitParallelState;
	function.parallel_function = GGParallelFunction;
	function.parallel_init = GGParallelInit;
	function.parallel_state_next = GGParallelStateNext;
	return function;
}

vector<int64_t> GGQueryInt64Column(ClientContext &context, const string &sql, const char *what) {
	Connection con(*context.db);
	auto result = con.Query(sql);
	if (!result->success) {
		throw BinderException(string(what) + " query failed: " + result->error);
	}
	vector<int64_t> out;
	for (idx_t r = 0; r < result->collection.Count(); r++) {
		auto v = result->GetValue(0, r);
		if (!v.is_null) {
			out.push_back(v.GetValue<int64_t>());
		}
	}
	return out;
}

//! gg_same_neighbour_paths(vertices_sql, sources_sql, path_table, path_src, path_dst,
//!                         filter_table, filter_src, filter_dst, hops) -> (w, v0..v{hops})
static unique_ptr<FunctionData> FilteredPathsBind(ClientContext &context, vector<Value> &inputs,
                                                  unordered_map<string, Value> &named_parameters,
                                                  vector<LogicalType> &input_table_types,
                                                  vector<string> &input_table_names,
                                                  vector<LogicalType> &return_types, vector<string> &names) {
	const auto hops = inputs[8].GetValue<int64_t>();
	if (hops < 1 || hops + 1 > GG_MAX_HOPS) {
		throw BinderException("gg_same_neighbour_paths: need 1 <= hops <= " + to_string(GG_MAX_HOPS - 1));
	}
	const string vertices_sql = inputs[0].ToString(), sources_sql = inputs[1].ToString();
	const string path_table = inputs[2].ToString(), path_src = inputs[3].ToString(), path_dst = inputs[4].ToString();
	const string filter_table = inputs[5].ToString(), filter_src = inputs[6].ToString(),
	             filter_dst = inputs[7].ToString();
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &ctx, GGOpened &opened) {
		opened.graph = make_shared<GGGraph>(0);
		PhysicalGGVertexSink vsink(opened.graph, {LogicalType::BIGINT}, 0);
		GGRunSinkPipeline(ctx, Statement(vertices_sql), vsink);
		PhysicalGGEdgeSink psink(opened.graph, {LogicalType::BIGINT, LogicalType::BIGINT}, 0);
		GGRunSinkPipeline(ctx, GGTableSource(ctx, path_table, {path_src, path_dst}, false), psink);
		PhysicalGGEdgeSink fsink(opened.graph, {LogicalType::BIGINT, LogicalType::BIGINT}, 0, true);
		GGRunSinkPipeline(ctx, GGTableSource(ctx, filter_table, {filter_src, filter_dst}, false), fsink);
		auto sources = GGQueryInt64Column(ctx, sources_sql, "gg_same_neighbour_paths: sources");
		opened.source = make_unique<PhysicalGGFilteredPaths>(opened.graph, (int)hops, move(sources), 0);
	};
	return_types = vector<LogicalType>(hops + 2, LogicalType::BIGINT);
	names.push_back("w");
	for (int64_t c = 0; c <= hops; c++) {
		names.push_back("v" + to_string(c));
	}
	return move(data);
}

static void CheckHops(int64_t k_min, int64_t k_max) {
	if (k_min < 1 || k_max < k_min || k_max > GG_MAX_HOPS) {
		throw BinderException("gg: need 1 <= k_min <= k_max <= " + to_string(GG_MAX_HOPS));
	}
}

static unique_ptr<FunctionData> KhopBindInternal(ClientContext &context, vector<Value> &inputs,
                                                 vector<LogicalType> &return_types, vector<string> &names,
                                                 bool count_only) {
	const auto k_min = inputs[5].GetValue<int64_t>(), k_max = inputs[6].GetValue<int64_t>();
	CheckHops(k_min, k_max);
	const GraphArguments graph(inputs);
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &ctx, GGOpened &opened) {
		opened.graph = GGBuildGraph(ctx, graph.Resolve(ctx));
		opened.source = make_unique<PhysicalGGPathExpand>(opened.graph, (int)k_min, (int)k_max, count_only,
		                                                  vector<int64_t>(), true, 0);
	};
	data->parallel_result = !count_only;
	return_types = PhysicalGGPathExpand::OutputTypes((int)k_max, count_only);
	names.push_back("hops");
	if (count_only) {
		names.push_back("rows");
		names.push_back("digest");
		names.push_back("traversed_edges");
	} else {
		for (int64_t c = 0; c <= k_max; c++) {
			names.push_back("v" + to_string(c));
		}
	}
	return move(data);
}

static unique_ptr<FunctionData> KhopBind(ClientContext &context, vector<Value> &inputs,
                                         unordered_map<string, Value> &named_parameters,
                                         vector<LogicalType> &input_table_types, vector<string> &input_table_names,
                                         vector<LogicalType> &return_types, vector<string> &names) {
	return KhopBindInternal(context, inputs, return_types, names, false);
}

static unique_ptr<FunctionData> KhopCountBind(ClientContext &context, vector<Value> &inputs,
                                              unordered_map<string, Value> &named_parameters,
                                              vector<LogicalType> &input_table_types,
                                              vector<string> &input_table_names, vector<LogicalType> &return_types,
                                              vector<string> &names) {
	return KhopBindInternal(context, inputs, return_types, names, true);
}

static unique_ptr<FunctionData> ShortestBind(ClientContext &context, vector<Value> &inputs,
                                             unordered_map<string, Value> &named_parameters,
                                             vector<LogicalType> &input_table_types,
                                             vector<string> &input_table_names, vector<LogicalType> &return_types,
                                             vector<string> &names) {
	const GraphArguments graph(inputs);
	const string sources_sql = inputs[5].ToString();
	const auto max_hops = inputs[6].GetValue<int64_t>();
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &ctx, GGOpened &opened) {
		opened.graph = GGBuildGraph(ctx, graph.Resolve(ctx));
		auto sources = GGQueryInt64Column(ctx, sources_sql, "gg_shortest_path: sources");
		opened.source = make_unique<PhysicalGGShortestPath>(opened.graph, move(sources), (int)max_hops, 0);
	};
	data->parallel_result = true;
	return_types = {LogicalType::BIGINT, LogicalType::BIGINT, LogicalType::INTEGER};
	names = {"startPerson", "friend", "hopCount"};
	return move(data);
}

static void LoadInternal(DatabaseInstance &db) {
	const vector<LogicalType> graph_args = {LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::VARCHAR,
	                                        LogicalType::VARCHAR, LogicalType::VARCHAR};
	auto khop_args = graph_args;
	khop_args.push_back(LogicalType::BIGINT);
	khop_args.push_back(LogicalType::BIGINT);
	auto sp_args = graph_args;
	sp_args.push_back(LogicalType::VARCHAR);
	sp_args.push_back(LogicalType::BIGINT);

	auto khop = GGScanFunction("gg_khop", khop_args, KhopBind);
	auto khop_count = GGScanFunction("gg_khop_count", khop_args, KhopCountBind);
	auto shortest = GGScanFunction("gg_shortest_path", sp_args, ShortestBind);
	auto filtered = GGScanFunction("gg_same_neighbour_paths",
	                               {LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::VARCHAR,
	                                LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::VARCHAR,
	                                LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::BIGINT},
	                               FilteredPathsBind);
	CreateTableFunctionInfo khop_info(khop), khop_count_info(khop_count), shortest_info(shortest),
	    filtered_info(filtered);

	Connection con(db);
	con.BeginTransaction();
	auto &catalog = Catalog::GetCatalog(*con.context);
	catalog.CreateTableFunction(*con.context, &khop_info);
	catalog.CreateTableFunction(*con.context, &khop_count_info);
	catalog.CreateTableFunction(*con.context, &shortest_info);
	catalog.CreateTableFunction(*con.context, &filtered_info);
	GGRegisterPlanRules(*con.context);
	con.Commit();
}

} // namespace duckdb

extern "C" {

void gg_duckdb_init(duckdb::DatabaseInstance &db) {
	duckdb::LoadInternal(db);
}

const char *gg_duckdb_version() {
	return duckdb::DuckDB::LibraryVersion();
}
}
