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
// DataChunk) -> Combine -> Finalize, then GetData until an empty chunk.  The true plan-level
// substitution (a PhysicalPlanGenerator rule + one BuildPipelines case) is described in INTEGRATION.md.
#include "duckdb.hpp"
#include "duckdb/catalog/catalog.hpp"
#include "duckdb/common/exception.hpp"
#include "duckdb/function/table_function.hpp"
#include "duckdb/main/client_context.hpp"
#include "duckdb/main/connection.hpp"
#include "duckdb/parallel/event.hpp"
#include "duckdb/parallel/pipeline.hpp"
#include "duckdb/parallel/thread_context.hpp"
#include "duckdb/parser/parsed_data/create_table_function_info.hpp"
#include "gg_operators.hpp"

namespace duckdb {

class GGNoopEvent : public Event {
public:
	explicit GGNoopEvent(Executor &executor) : Event(executor) {
	}
	void Schedule() override {
	}
};

static string Quote(const string &ident) {
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

//! Run `sql` on a side connection and push its chunks through `sink` the way a pipeline would.
static void RunSinkPipeline(ClientContext &context, const string &sql, PhysicalOperator &sink) {
	Connection con(*context.db);
	auto result = con.SendQuery(sql);
	if (!result->success) {
		throw BinderException("gg: scanning the base table failed: " + result->error);
	}
	ThreadContext thread(context);
	ExecutionContext ec(context, thread);
	sink.sink_state = sink.GetGlobalSinkState(context);
	auto lstate = sink.GetLocalSinkState(ec);
	while (true) {
		auto chunk = result->Fetch();
		if (!chunk || chunk->size() == 0) {
			break;
		}
		sink.Sink(ec, *sink.sink_state, *lstate, *chunk);
	}
	sink.Combine(ec, *sink.sink_state, *lstate);
	Pipeline pipeline(context.executor);
	GGNoopEvent event(context.executor);
	sink.Finalize(pipeline, event, context, *sink.sink_state);
}

static shared_ptr<GGGraph> BuildGraph(ClientContext &context, vector<Value> &inputs) {
	auto graph = make_shared<GGGraph>(0);
	const string vt = inputs[0].ToString(), vk = inputs[1].ToString(), et = inputs[2].ToString(),
	             es = inputs[3].ToString(), ed = inputs[4].ToString();
	PhysicalGGVertexSink vsink(graph, {LogicalType::BIGINT}, 0);
	RunSinkPipeline(context, "SELECT " + Quote(vk) + " FROM " + Quote(vt), vsink);
	// rowid comes out of the scan as a sequence vector (row_group.cpp:335): the sink Orrifies it
	PhysicalGGEdgeSink esink(graph, {LogicalType::BIGINT, LogicalType::BIGINT, LogicalType::BIGINT}, 0);
	RunSinkPipeline(context, "SELECT " + Quote(es) + ", " + Quote(ed) + ", rowid FROM " + Quote(et), esink);
	return graph;
}

//! Bind data: the source operator plus its state; the function call is GetData.
struct GGFunctionData : public TableFunctionData {
	shared_ptr<GGGraph> graph;
	unique_ptr<PhysicalOperator> source;
	unique_ptr<GlobalSourceState> gstate;
};

struct GGOperatorData : public FunctionOperatorData {};

static unique_ptr<FunctionOperatorData> GGInit(ClientContext &context, const FunctionData *bind_data,
                                               const vector<column_t> &column_ids, TableFilterCollection *filters) {
	return make_unique<GGOperatorData>();
}

static void GGFunction(ClientContext &context, const FunctionData *bind_data_p, FunctionOperatorData *operator_state,
                       DataChunk *input, DataChunk &output) {
	auto &data = (GGFunctionData &)*bind_data_p;
	ThreadContext thread(context);
	ExecutionContext ec(context, thread);
	LocalSourceState lstate;
	data.source->GetData(ec, output, *data.gstate, lstate);
}

static vector<int64_t> QueryInt64Column(ClientContext &context, const string &sql, const char *what) {
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
	auto data = make_unique<GGFunctionData>();
	data->graph = make_shared<GGGraph>(0);
	PhysicalGGVertexSink vsink(data->graph, {LogicalType::BIGINT}, 0);
	RunSinkPipeline(context, inputs[0].ToString(), vsink);
	PhysicalGGEdgeSink psink(data->graph, {LogicalType::BIGINT, LogicalType::BIGINT}, 0);
	RunSinkPipeline(context,
	                "SELECT " + Quote(inputs[3].ToString()) + ", " + Quote(inputs[4].ToString()) + " FROM " +
	                    Quote(inputs[2].ToString()),
	                psink);
	PhysicalGGEdgeSink fsink(data->graph, {LogicalType::BIGINT, LogicalType::BIGINT}, 0, true);
	RunSinkPipeline(context,
	                "SELECT " + Quote(inputs[6].ToString()) + ", " + Quote(inputs[7].ToString()) + " FROM " +
	                    Quote(inputs[5].ToString()),
	                fsink);
	auto sources = QueryInt64Column(context, inputs[1].ToString(), "gg_same_neighbour_paths: sources");
	data->source = make_unique<PhysicalGGFilteredPaths>(data->graph, (int)hops, move(sources), 0);
	data->gstate = data->source->GetGlobalSourceState(context);
	return_types = data->source->GetTypes();
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
	auto data = make_unique<GGFunctionData>();
	data->graph = BuildGraph(context, inputs);
	data->source = make_unique<PhysicalGGPathExpand>(data->graph, (int)k_min, (int)k_max, count_only,
	                                                 vector<int64_t>(), true, 0);
	data->gstate = data->source->GetGlobalSourceState(context);
	return_types = data->source->GetTypes();
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
	auto data = make_unique<GGFunctionData>();
	data->graph = BuildGraph(context, inputs);
	auto sources = QueryInt64Column(context, inputs[5].ToString(), "gg_shortest_path: sources");
	data->source = make_unique<PhysicalGGShortestPath>(data->graph, move(sources), (int)inputs[6].GetValue<int64_t>(), 0);
	data->gstate = data->source->GetGlobalSourceState(context);
	return_types = data->source->GetTypes();
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

	TableFunction khop("gg_khop", khop_args, GGFunction, KhopBind, GGInit);
	TableFunction khop_count("gg_khop_count", khop_args, GGFunction, KhopCountBind, GGInit);
	TableFunction shortest("gg_shortest_path", sp_args, GGFunction, ShortestBind, GGInit);
	TableFunction filtered("gg_same_neighbour_paths",
	                       {LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::VARCHAR,
	                        LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::VARCHAR,
	                        LogicalType::BIGINT},
	                       GGFunction, FilteredPathsBind, GGInit);
	CreateTableFunctionInfo khop_info(khop), khop_count_info(khop_count), shortest_info(shortest),
	    filtered_info(filtered);

	Connection con(db);
	con.BeginTransaction();
	auto &catalog = Catalog::GetCatalog(*con.context);
	catalog.CreateTableFunction(*con.context, &khop_info);
	catalog.CreateTableFunction(*con.context, &khop_count_info);
	catalog.CreateTableFunction(*con.context, &shortest_info);
	catalog.CreateTableFunction(*con.context, &filtered_info);
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
