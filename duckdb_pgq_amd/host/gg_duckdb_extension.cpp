// gg_duckdb_extension.cpp — zero-patch SQL surface for the GPU graph operators.
//
// Loaded with   LOAD '<repo>/duckdb_pgq_amd/gg_duckdb.duckdb_extension';
// (PhysicalLoad: dlopen + <basename>_init / <basename>_version,
//  src/execution/operator/helper/physical_load.cpp:29-70 of the reference).  It registers table
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
//   gg_graph_pin(vertex_table, vertex_key, edge_table, src_col, dst_col) / gg_graph_unpin() / gg_graph_pins()
//        -> keep that graph on the device for later statements (a snapshot; see "Pinned graphs" below)
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
#include "duckdb/catalog/catalog_entry/table_catalog_entry.hpp"
#include "duckdb/storage/data_table.hpp"
#include "duckdb/common/exception.hpp"
#include "duckdb/function/table_function.hpp"
#include "duckdb/main/client_context.hpp"
#include "duckdb/main/connection.hpp"
#include "duckdb/parallel/parallel_state.hpp"
#include "duckdb/parallel/thread_context.hpp"
#include "duckdb/parser/parsed_data/create_table_function_info.hpp"
#include "duckdb/transaction/transaction.hpp"
#include "gg_extension.hpp"

#include <csignal>
#include <execinfo.h>
#include <unistd.h>

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

//===--------------------------------------------------------------------===//
// Pinned graphs
//===--------------------------------------------------------------------===//
// Every statement reads its base tables again (like the hash-join builds it replaces), which makes a
// selective query — one source, two hops — pay for the whole edge table.  `gg_graph_pin(...)` builds the
// graph of a (vertex table, key, edge table, src, dst) combination once and keeps it on the device, the way
// a property graph or an index is declared once.  A pinned graph is a SNAPSHOT of what the pinning
// transaction saw, and this version of the reference has no per-table modification counter to hang an
// invalidation on, so its use is hedged:
//   - nothing uses a pinned graph unless the connection asked for it: PRAGMA gg_use_pinned_graphs (per
//     connection, off by default; PRAGMA gg_ignore_pinned_graphs turns it off again).  Without it planner
//     rules and table functions read the tables, in the statement's own transaction, like the joins they replace;
//   - a transaction that has changed anything (Transaction::ChangesMade: local appends, updates, deletes)
//     neither creates nor uses pinned graphs;
//   - an INSERT, DELETE or UPDATE planned OR executed on a pinned table — by any connection of this process,
//     committed or not — drops the pins on it (gg_plan_hook.c observes the three CreatePlan overloads, and the
//     BuildPipelines rule of gg_pipeline.cpp sees every plan the executor is about to run, so a statement
//     prepared before the pin drops it when it is executed; without the shim there are no planner rules either,
//     only the table functions);
//   - a changed row count (appends by other means) drops the pin as well.
// What remains is the caller's business and is documented in INTEGRATION.md: a graph pinned while another
// connection holds an uncommitted change does not see that change when it commits — unpin before such work.
struct PinnedGraph {
	idx_t vertex_oid, edge_oid; // catalog oids (never reused), vertex_oid = 0: vertex set = endpoint ids
	column_t vertex_key, src, dst;
	idx_t vertex_rows, edge_rows;
	shared_ptr<GGGraph> graph;
};
static mutex g_pinned_lock;
static vector<PinnedGraph> g_pinned;

static bool PinKey(const GGGraphSpec &spec, PinnedGraph &key) {
	if (!spec.edges.table || spec.edges.columns.size() != 2 || (!spec.vertices.Empty() && !spec.vertices.table)) {
		return false; // only plain tables, and no rowid payload
	}
	key.edge_oid = spec.edges.table->oid;
	key.src = spec.edges.columns[0];
	key.dst = spec.edges.columns[1];
	key.edge_rows = spec.edges.table->storage->GetTotalRows();
	key.vertex_oid = spec.vertices.table ? spec.vertices.table->oid : 0;
	key.vertex_key = spec.vertices.table ? spec.vertices.columns[0] : 0;
	key.vertex_rows = spec.vertices.table ? spec.vertices.table->storage->GetTotalRows() : 0;
	return true;
}

static shared_ptr<GGGraph> FindPinned(ClientContext &context, const GGGraphSpec &spec) {
	PinnedGraph key;
	if (!GGGetConnectionFlags(context).pinned_graphs || Transaction::GetTransaction(context).ChangesMade() ||
	    !PinKey(spec, key)) {
		return nullptr;
	}
	lock_guard<mutex> guard(g_pinned_lock);
	for (idx_t i = 0; i < g_pinned.size(); i++) {
		auto &p = g_pinned[i];
		if (p.vertex_oid == key.vertex_oid && p.edge_oid == key.edge_oid && p.vertex_key == key.vertex_key &&
		    p.src == key.src && p.dst == key.dst) {
			if (p.vertex_rows == key.vertex_rows && p.edge_rows == key.edge_rows) {
				return p.graph;
			}
			g_pinned.erase(g_pinned.begin() + i); // rows were appended (or the table rebuilt): stale
			return nullptr;
		}
	}
	return nullptr;
}

static shared_ptr<GGGraph> BuildGraphNow(ClientContext &context, const GGGraphSpec &spec) {
	PhaseTimer timer;
	const bool edge_rowids = spec.edges.columns.size() >= 3 || (!spec.edges.table && spec.edges_with_rowid);
	// (a third edge column is the rowid: walks with their edges)
	auto graph = make_shared<GGGraph>(0, edge_rowids, edge_rowids ? 1 : spec.shards);
	timer.Lap("device context");
	const bool derive = spec.vertices.Empty();
	if (!derive) {
		PhysicalGGVertexSink vsink(graph, {LogicalType::BIGINT}, 0);
		GGRunSinkPipeline(context, spec.vertices, vsink);
		timer.Lap("vertex table ingest");
	}
	vector<LogicalType> edge_types(edge_rowids ? 3 : 2, LogicalType::BIGINT);
	PhysicalGGEdgeSink esink(graph, edge_types, 0, false, derive);
	GGRunSinkPipeline(context, spec.edges, esink);
	timer.Lap("edge ingest + CSR build");
	return graph;
}

shared_ptr<GGGraph> GGBuildGraph(ClientContext &context, const GGGraphSpec &spec) {
	if (auto pinned = FindPinned(context, spec)) {
		return pinned;
	}
	return BuildGraphNow(context, spec);
}

void GGDropPinsOfTable(idx_t table_oid) {
	lock_guard<mutex> guard(g_pinned_lock);
	for (idx_t i = 0; i < g_pinned.size();) {
		if (g_pinned[i].edge_oid == table_oid || g_pinned[i].vertex_oid == table_oid) {
			g_pinned.erase(g_pinned.begin() + i);
		} else {
			i++;
		}
	}
}

// ---- per-connection switches
static mutex g_flags_lock;
static vector<std::pair<weak_ptr<ClientContext>, GGConnectionFlags>> g_flags;

static bool GGDefaultRules() {
	static const bool on = [] {
		auto env = std::getenv("GG_PLAN_RULE"); // process-wide default for connections that never said otherwise
		return env && env[0] == '1';
	}();
	return on;
}

GGConnectionFlags GGGetConnectionFlags(ClientContext &context) {
	lock_guard<mutex> guard(g_flags_lock);
	for (auto &entry : g_flags) {
		auto owner = entry.first.lock();
		if (owner.get() == &context) {
			return entry.second;
		}
	}
	GGConnectionFlags flags;
	flags.rules = GGDefaultRules();
	return flags;
}

void GGSetConnectionFlags(ClientContext &context, const GGConnectionFlags &flags) {
	lock_guard<mutex> guard(g_flags_lock);
	for (idx_t i = 0; i < g_flags.size();) { // connections that are gone take their entries with them
		if (g_flags[i].first.expired()) {
			g_flags.erase(g_flags.begin() + i);
		} else {
			i++;
		}
	}
	for (auto &entry : g_flags) {
		if (entry.first.lock().get() == &context) {
			entry.second = flags;
			return;
		}
	}
	g_flags.emplace_back(weak_ptr<ClientContext>(context.shared_from_this()), flags);
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
	// the threads that drain the result also run the rest of the pipeline (filters, aggregates above the
	// scan), so more of them than PCIe alone needs: GG_SCAN_THREADS, default 32
	static const idx_t threads = [] {
		auto env = std::getenv("GG_SCAN_THREADS");
		const idx_t n = env ? (idx_t)std::strtoull(env, nullptr, 10) : 32;
		return MaxValue<idx_t>(n, 2);
	}();
	return ((const GGFunctionData &)*bind_data).parallel_result ? threads : 1;
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

//! gg_graph_pin(vertex_table, vertex_key, edge_table, src_col, dst_col) -> (vertices, edges, build_ms);
//! vertex_table = '' pins the edge-only form (vertex set = endpoint ids).  gg_graph_unpin() drops all.
struct PinResultData : public TableFunctionData {
	int64_t vertices = 0, edges = 0;
	double build_ms = 0;
	bool done = false;
};

static unique_ptr<FunctionData> GraphPinBind(ClientContext &context, vector<Value> &inputs,
                                             unordered_map<string, Value> &named_parameters,
                                             vector<LogicalType> &input_table_types, vector<string> &input_table_names,
                                             vector<LogicalType> &return_types, vector<string> &names) {
	const string vertex_table = inputs[0].ToString();
	GGGraphSpec spec;
	if (!vertex_table.empty()) {
		spec.vertices = GGTableSource(context, vertex_table, {inputs[1].ToString()}, false);
	}
	spec.edges = GGTableSource(context, inputs[2].ToString(), {inputs[3].ToString(), inputs[4].ToString()}, false);
	PinnedGraph pin;
	if (!PinKey(spec, pin)) {
		throw BinderException("gg_graph_pin: only base tables can be pinned");
	}
	if (Transaction::GetTransaction(context).ChangesMade()) {
		throw BinderException("gg_graph_pin: this transaction has uncommitted changes; a pinned graph would keep "
		                      "them whether or not they commit — commit or roll back first");
	}
	auto t0 = std::chrono::steady_clock::now();
	pin.graph = BuildGraphNow(context, spec);
	auto result = make_unique<PinResultData>();
	result->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	uint64_t v = 0, e = 0;
	GGGraph::Check(gg_csr_info(pin.graph->csr, &v, &e, nullptr), "gg_csr_info");
	result->vertices = (int64_t)v;
	result->edges = (int64_t)e;
	{
		lock_guard<mutex> guard(g_pinned_lock);
		for (idx_t i = 0; i < g_pinned.size(); i++) { // replace an older pin of the same combination
			auto &p = g_pinned[i];
			if (p.vertex_oid == pin.vertex_oid && p.edge_oid == pin.edge_oid && p.vertex_key == pin.vertex_key &&
			    p.src == pin.src && p.dst == pin.dst) {
				g_pinned.erase(g_pinned.begin() + i);
				break;
			}
		}
		g_pinned.push_back(move(pin));
	}
	return_types = {LogicalType::BIGINT, LogicalType::BIGINT, LogicalType::DOUBLE};
	names = {"vertices", "edges", "build_ms"};
	return move(result);
}

static void GraphPinFunction(ClientContext &context, const FunctionData *bind_data, FunctionOperatorData *operator_state,
                             DataChunk *input, DataChunk &output) {
	auto &data = (PinResultData &)*bind_data;
	if (data.done) {
		return;
	}
	data.done = true;
	output.SetValue(0, 0, Value::BIGINT(data.vertices));
	output.SetValue(1, 0, Value::BIGINT(data.edges));
	output.SetValue(2, 0, Value::DOUBLE(data.build_ms));
	output.SetCardinality(1);
}

static unique_ptr<FunctionData> GraphUnpinBind(ClientContext &context, vector<Value> &inputs,
                                               unordered_map<string, Value> &named_parameters,
                                               vector<LogicalType> &input_table_types,
                                               vector<string> &input_table_names, vector<LogicalType> &return_types,
                                               vector<string> &names) {
	auto result = make_unique<PinResultData>();
	{
		lock_guard<mutex> guard(g_pinned_lock);
		result->vertices = (int64_t)g_pinned.size();
		g_pinned.clear();
	}
	return_types = {LogicalType::BIGINT};
	names = {"unpinned"};
	return move(result);
}

static void GraphUnpinFunction(ClientContext &context, const FunctionData *bind_data, FunctionOperatorData *operator_state,
                               DataChunk *input, DataChunk &output) {
	auto &data = (PinResultData &)*bind_data;
	if (data.done) {
		return;
	}
	data.done = true;
	output.SetValue(0, 0, Value::BIGINT(data.vertices));
	output.SetCardinality(1);
}

//! gg_graph_pins() -> number of graphs currently pinned (tests; monitoring)
static unique_ptr<FunctionData> GraphPinsBind(ClientContext &context, vector<Value> &inputs,
                                              unordered_map<string, Value> &named_parameters,
                                              vector<LogicalType> &input_table_types, vector<string> &input_table_names,
                                              vector<LogicalType> &return_types, vector<string> &names) {
	auto result = make_unique<PinResultData>();
	{
		lock_guard<mutex> guard(g_pinned_lock);
		result->vertices = (int64_t)g_pinned.size();
	}
	return_types = {LogicalType::BIGINT};
	names = {"pinned"};
	return move(result);
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
	TableFunction pin("gg_graph_pin", graph_args, GraphPinFunction, GraphPinBind);
	TableFunction unpin("gg_graph_unpin", {}, GraphUnpinFunction, GraphUnpinBind);
	TableFunction pins("gg_graph_pins", {}, GraphUnpinFunction, GraphPinsBind);
	CreateTableFunctionInfo khop_info(khop), khop_count_info(khop_count), shortest_info(shortest),
	    filtered_info(filtered), pin_info(pin), unpin_info(unpin), pins_info(pins);

	Connection con(db);
	con.BeginTransaction();
	auto &catalog = Catalog::GetCatalog(*con.context);
	catalog.CreateTableFunction(*con.context, &khop_info);
	catalog.CreateTableFunction(*con.context, &khop_count_info);
	catalog.CreateTableFunction(*con.context, &shortest_info);
	catalog.CreateTableFunction(*con.context, &filtered_info);
	catalog.CreateTableFunction(*con.context, &pin_info);
	catalog.CreateTableFunction(*con.context, &unpin_info);
	catalog.CreateTableFunction(*con.context, &pins_info);
	GGRegisterPlanRules(*con.context);
	con.Commit();
}

} // namespace duckdb

// GG_CRASH_TRACE=1: a backtrace on stderr when the process dies of SIGSEGV / SIGBUS / SIGABRT (diagnostic: the boxes
// this runs on have no debugger).  Installed with sigaction and SA_RESETHAND (the handler runs once, then the default
// action or whatever the application had takes over); the handlers that were installed before are kept and chained to;
// backtrace() is called once at load, so that its lazy dlopen of libgcc (which allocates) does not happen inside a signal
// handler.  A diagnostic all the same: off unless the variable is set.
static struct sigaction g_previous_action[3];
static const int g_crash_signals[3] = {SIGSEGV, SIGBUS, SIGABRT};

static void GGCrashHandler(int sig, siginfo_t *info, void *ucontext) {
	void *frames[64];
	const int n = backtrace(frames, 64);
	const char msg[] = "\n=== gg crash trace ===\n";
	(void)!write(2, msg, sizeof(msg) - 1);
	backtrace_symbols_fd(frames, n, 2);
	for (int i = 0; i < 3; i++) {
		if (g_crash_signals[i] != sig) {
			continue;
		}
		auto &old = g_previous_action[i];
		if ((old.sa_flags & SA_SIGINFO) && old.sa_sigaction) {
			old.sa_sigaction(sig, info, ucontext); // the embedding application's own handler
			return;
		}
		if (old.sa_handler != SIG_DFL && old.sa_handler != SIG_IGN && old.sa_handler) {
			old.sa_handler(sig);
			return;
		}
	}
	// (SA_RESETHAND put the default action back: re-raising ends the process the way it would have ended)
	raise(sig);
}

extern "C" {

void gg_duckdb_init(duckdb::DatabaseInstance &db) {
	if (std::getenv("GG_CRASH_TRACE")) {
		void *warm[2];
		(void)backtrace(warm, 2); // (loads libgcc's unwinder now, not in the handler)
		for (int i = 0; i < 3; i++) {
			struct sigaction action;
			memset(&action, 0, sizeof(action));
			action.sa_sigaction = GGCrashHandler;
			action.sa_flags = SA_SIGINFO | SA_RESETHAND;
			sigemptyset(&action.sa_mask);
			sigaction(g_crash_signals[i], &action, &g_previous_action[i]);
		}
	}
	duckdb::LoadInternal(db);
}

const char *gg_duckdb_version() {
	return duckdb::DuckDB::LibraryVersion();
}
}
