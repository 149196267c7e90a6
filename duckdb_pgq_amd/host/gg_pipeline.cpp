// gg_pipeline.cpp — the GPU graph's build side as REAL pipeline sinks of the reference's executor.
//
// The reference schedules a hash join's build side as a child pipeline whose sink is the join:
// Executor::BuildPipelines (src/parallel/executor.cpp:385-580) creates a Pipeline per sink operator, makes the
// consuming pipeline depend on it, and PipelineExecutor drives Sink / Combine / Finalize on the worker threads
// (src/parallel/pipeline_executor.cpp).  Its switch over PhysicalOperatorType has no case for operators it does not
// know ("Unimplemented sink type!", executor.cpp:475; "Operator not supported yet" for a non-sink with several
// children, :571).  PhysicalGGGraphScan is such an operator: a SOURCE (the walks, counts, endpoint sets the GPU
// produces) whose children are SINKS (the vertex / edge tables flowing into the device graph):
//
//     GG_PATH_EXPAND                       <- source of the consuming pipeline
//       GG_VERTEX_SINK <- SEQ_SCAN person  <- child pipeline 1
//       GG_EDGE_SINK   <- SEQ_SCAN knows   <- child pipeline 2 (depends on 1: its Finalize builds the CSR)
//
// GGBuildPipelinesRule is the missing case.  The maintainers' route is four lines in BuildPipelines' non-sink branch
// (INTEGRATION.md §3); the reference tree is read-only here, so the interposition shim (gg_plan_hook.c) offers every
// BuildPipelines call to this rule first, exactly as it does for the CreatePlan rules.  With it the table scans are
// the reference's own PhysicalTableScan, parallelised and profiled by its executor, and EXPLAIN ANALYZE attributes
// scan, sink and GPU time to separate operators.
#include <cstdlib>
#include <dlfcn.h>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "duckdb.hpp"
#include "duckdb/execution/operator/scan/physical_table_scan.hpp"
#include "duckdb/function/table/table_scan.hpp"
#include "duckdb/catalog/catalog_entry/table_catalog_entry.hpp"
#include "duckdb/storage/data_table.hpp"
// Executor::BuildPipelines and the pipeline lists it fills are private; the rule below is the body of a case the
// reference's own member function would hold.  Nothing else in this file touches non-public members — and with
// -DGG_REFERENCE_CALLOUTS (an extension built only for a reference that carries oracle/callout.patch, where that
// case is the executor's own code) neither the rule nor this access is compiled at all.
#ifndef GG_REFERENCE_CALLOUTS
#define private public
#include "duckdb/execution/executor.hpp"
#include "duckdb/parallel/pipeline.hpp"
#include "duckdb/execution/operator/set/physical_recursive_cte.hpp"
#undef private
#endif
#include "duckdb/execution/operator/helper/physical_execute.hpp"
#include "duckdb/execution/operator/join/physical_delim_join.hpp"
#include "duckdb/execution/operator/persistent/physical_delete.hpp"
#include "duckdb/execution/operator/persistent/physical_insert.hpp"
#include "duckdb/execution/operator/persistent/physical_update.hpp"

#include "gg_extension.hpp"
#include "gg_operators.hpp"
#include "gg_pipeline.hpp"
#include "gg_plan_hook.h"

namespace duckdb {

//===--------------------------------------------------------------------===//
// Lazy sinks: the device context is created when the pipeline starts, never at EXPLAIN / PREPARE
//===--------------------------------------------------------------------===//
PhysicalGGLazySink::PhysicalGGLazySink(shared_ptr<GGGraphSlot> slot_p, Kind kind_p, vector<LogicalType> types,
                                       idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::INVALID, move(types), estimated_cardinality), slot(move(slot_p)),
      kind(kind_p) {
}

PhysicalGGLazySink::PhysicalGGLazySink(shared_ptr<GGGraphSlot> slot_p, EdgeOptions options_p, vector<LogicalType> types,
                                       idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::INVALID, move(types), estimated_cardinality), slot(move(slot_p)),
      kind(EDGES_CUSTOM), options(options_p) {
}

unique_ptr<GlobalSinkState> PhysicalGGLazySink::GetGlobalSinkState(ClientContext &context) const {
	lock_guard<mutex> guard(slot->lock);
	if (kind == VERTICES || kind == EDGES_DERIVE_VERTICES || (kind == EDGES_CUSTOM && options.first)) {
		slot->graph = make_shared<GGGraph>(0, false, slot->shards); // first sink of an execution: a fresh graph
	}
	if (!slot->graph) {
		throw InternalException("GG_EDGE_SINK scheduled before its vertex sink");
	}
	if (kind == EDGES_CUSTOM) {
		inner = make_unique<PhysicalGGEdgeSink>(slot->graph, types, estimated_cardinality, options.as_filter,
		                                        options.derive_vertices, options.keep_vertices, options.build);
	} else if (kind == VERTICES) {
		inner = make_unique<PhysicalGGVertexSink>(slot->graph, types, estimated_cardinality);
	} else {
		inner = make_unique<PhysicalGGEdgeSink>(slot->graph, types, estimated_cardinality, false,
		                                        kind == EDGES_DERIVE_VERTICES);
	}
	return inner->GetGlobalSinkState(context);
}

unique_ptr<LocalSinkState> PhysicalGGLazySink::GetLocalSinkState(ExecutionContext &context) const {
	return inner->GetLocalSinkState(context);
}

SinkResultType PhysicalGGLazySink::Sink(ExecutionContext &context, GlobalSinkState &gstate, LocalSinkState &lstate,
                                        DataChunk &input) const {
	return inner->Sink(context, gstate, lstate, input);
}

void PhysicalGGLazySink::Combine(ExecutionContext &context, GlobalSinkState &gstate, LocalSinkState &lstate) const {
	inner->Combine(context, gstate, lstate);
}

SinkFinalizeType PhysicalGGLazySink::Finalize(Pipeline &pipeline, Event &event, ClientContext &context,
                                              GlobalSinkState &gstate) const {
	auto result = inner->Finalize(pipeline, event, context, gstate);
	if (kind == EDGES_CUSTOM && options.clear_edges_after) {
		lock_guard<mutex> guard(slot->lock);
		GGGraph::Check(gg_staging_clear_edges(slot->graph->ctx), "gg_staging_clear_edges");
	}
	return result;
}

string PhysicalGGLazySink::GetName() const {
	return kind == VERTICES ? "GG_VERTEX_SINK" : "GG_EDGE_SINK";
}

//===--------------------------------------------------------------------===//
// The scan: sinks below, a GG source inside
//===--------------------------------------------------------------------===//
namespace {
//! Pipeline::Ready creates every pipeline's source state when the pipelines are SCHEDULED (executor.cpp:46-191),
//! before any of them has run — the graph does not exist yet.  The GG source inside is therefore made on first
//! use: MaxThreads (asked when the consuming pipeline is launched, after its dependencies), or the first GetData.
class GraphScanGlobalState : public GlobalSourceState {
public:
	GraphScanGlobalState(const PhysicalGGGraphScan &op, ClientContext &context)
	    : op(op), context(context), slot(op.slot), keep_graph(op.keep_graph) {
	}
	~GraphScanGlobalState() override {
		// the executor drops its pipelines (and this state with them) when the statement is done: the device graph goes
		// with it, a prepared plan does not sit on HBM between executions.  (The plan itself may already be gone here:
		// only what this state holds on its own is touched.)
		state.reset();
		source.reset();
		if (!keep_graph) {
			lock_guard<mutex> guard(slot->lock);
			slot->graph.reset();
		}
	}
	const PhysicalGGGraphScan &op;  // (valid while the plan runs: Ensure, MaxThreads)
	ClientContext &context;
	shared_ptr<GGGraphSlot> slot;
	bool keep_graph;
	mutex lock;
	unique_ptr<PhysicalOperator> source;
	unique_ptr<GlobalSourceState> state;

	void Ensure() {
		lock_guard<mutex> guard(lock);
		if (state) {
			return;
		}
		shared_ptr<GGGraph> graph;
		{
			lock_guard<mutex> slot_guard(op.slot->lock);
			graph = op.slot->graph;
		}
		if (!graph || !graph->csr) {
			throw InternalException(op.name + " scheduled before its sinks built the graph");
		}
		source = op.factory(context, graph);
		state = source->GetGlobalSourceState(context);
	}
	idx_t MaxThreads() override {
		Ensure();
		return state->MaxThreads();
	}
};
} // namespace

PhysicalGGGraphScan::PhysicalGGGraphScan(vector<LogicalType> types, string name_p, string description_p,
                                         shared_ptr<GGGraphSlot> slot_p, Factory factory_p, bool parallel_result_p,
                                         idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::INVALID, move(types), estimated_cardinality), name(move(name_p)),
      description(move(description_p)), slot(move(slot_p)), factory(move(factory_p)),
      parallel_result(parallel_result_p) {
}

unique_ptr<GlobalSourceState> PhysicalGGGraphScan::GetGlobalSourceState(ClientContext &context) const {
	return make_unique<GraphScanGlobalState>(*this, context);
}

unique_ptr<LocalSourceState> PhysicalGGGraphScan::GetLocalSourceState(ExecutionContext &context,
                                                                      GlobalSourceState &gstate_p) const {
	auto &gstate = (GraphScanGlobalState &)gstate_p;
	gstate.Ensure();
	return gstate.source->GetLocalSourceState(context, *gstate.state);
}

void PhysicalGGGraphScan::GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate_p,
                                  LocalSourceState &lstate) const {
	auto &gstate = (GraphScanGlobalState &)gstate_p;
	gstate.source->GetData(context, chunk, *gstate.state, lstate);
}

string PhysicalGGGraphScan::GetName() const {
	return name;
}

string PhysicalGGGraphScan::ParamsToString() const {
	return description;
}

//! Scan tasks of a pipeline that ends in a staging sink.  The reference's executor would put every thread it has on
//! the table scan (pipeline.cpp:91-118: min(source MaxThreads, scheduler threads)); the sink ends in ONE PCIe link,
//! which eight appenders saturate (scripts/bench_staging_native.cpp: 4-64 threads all within 47-51 GB/s of the
//! link's 57), and past that the threads only queue for the staging block: the SF100 `count(*)` statement took
//! 15.5 ms with PRAGMA threads = 8..32, 16.2 with 64, 19.5-23.8 with 256 (this host's default) — and a MEDIAN of 68 ms
//! there, 256 sleepers being woken per block.  So the scan offers at most GG_INGEST_TASKS tasks (default 8, the same
//! ceiling the scan-function route's ingest uses, gg_ingest.cpp) whatever PRAGMA threads says.
static idx_t GGStagingScanMaxThreads(ClientContext &context, const FunctionData *bind_data_p) {
	static const idx_t task_cap = [] {
		auto env = std::getenv("GG_INGEST_TASKS");
		const idx_t n = env ? (idx_t)std::strtoull(env, nullptr, 10) : 8;
		return MaxValue<idx_t>(1, n);
	}();
	auto &bind_data = (const TableScanBindData &)*bind_data_p;
	return MinValue<idx_t>(bind_data.table->storage->MaxThreads(context), task_cap); // (table_scan.cpp:81-85)
}

//! PhysicalTableScan of the given columns of a base table (what plan_get.cpp:47-60 builds for a seq_scan)
unique_ptr<PhysicalOperator> GGBaseTableScan(const GGScanSource &source) {
	auto &table = *source.table;
	vector<LogicalType> types;  // of the scanned columns
	vector<string> names;       // of ALL columns: PhysicalTableScan indexes them by column id (as LogicalGet::names)
	for (auto column : source.columns) {
		types.push_back(column == COLUMN_IDENTIFIER_ROW_ID ? LogicalType::BIGINT : table.columns[column].type);
	}
	for (auto &column : table.columns) {
		names.push_back(column.name);
	}
	auto bind = make_unique<TableScanBindData>(&table);
	auto function = TableScanFunction::GetFunction();
	function.max_threads = GGStagingScanMaxThreads;
	return make_unique<PhysicalTableScan>(move(types), move(function), move(bind), source.columns, move(names), nullptr,
	                                      table.storage->GetTotalRows());
}

bool GGPipelineSinksAvailable(ClientContext &context, const GGGraphSpec &spec) {
	// GG_NO_PIPELINE_SINKS=1: the scan-function route for every plan (the sinks are then driven from the scan's init)
	if (std::getenv("GG_NO_PIPELINE_SINKS") || !gg_pipeline_rule_registered() || !spec.edges.table ||
	    (!spec.vertices.Empty() && !spec.vertices.table)) {
		return false; // views and statements are read through a side connection: the scan-function route
	}
	// a connection that asked for pinned graphs may not need to read the tables at all: decided at run time there
	return !GGGetConnectionFlags(context).pinned_graphs;
}

unique_ptr<PhysicalOperator> GGMakeGraphScan(const vector<GGSinkSpec> &sinks, vector<LogicalType> types, string name,
                                             string description, bool parallel_result,
                                             PhysicalGGGraphScan::Factory factory, idx_t estimated_cardinality) {
	auto slot = make_shared<GGGraphSlot>();
	auto scan = make_unique<PhysicalGGGraphScan>(move(types), move(name), move(description), slot, move(factory),
	                                             parallel_result, estimated_cardinality);
	for (auto &spec : sinks) {
		auto rows = GGBaseTableScan(spec.rows);
		auto sink = make_unique<PhysicalGGLazySink>(slot, spec.options, rows->types, rows->estimated_cardinality);
		sink->children.push_back(move(rows));
		scan->children.push_back(move(sink));
	}
	return move(scan);
}

unique_ptr<PhysicalOperator> GGMakeGraphScan(const GGGraphSpec &spec, vector<LogicalType> types, string name,
                                             string description, bool parallel_result,
                                             PhysicalGGGraphScan::Factory factory, idx_t estimated_cardinality) {
	auto slot = make_shared<GGGraphSlot>();
	slot->shards = spec.shards;
	auto scan = make_unique<PhysicalGGGraphScan>(move(types), move(name), move(description), slot, move(factory),
	                                             parallel_result, estimated_cardinality);
	const bool derive = spec.vertices.Empty();
	if (!derive) {
		auto rows = GGBaseTableScan(spec.vertices);
		auto sink = make_unique<PhysicalGGLazySink>(slot, PhysicalGGLazySink::VERTICES, rows->types,
		                                            rows->estimated_cardinality);
		sink->children.push_back(move(rows));
		scan->children.push_back(move(sink));
	}
	auto rows = GGBaseTableScan(spec.edges);
	auto sink = make_unique<PhysicalGGLazySink>(
	    slot, derive ? PhysicalGGLazySink::EDGES_DERIVE_VERTICES : PhysicalGGLazySink::EDGES, rows->types,
	    rows->estimated_cardinality);
	sink->children.push_back(move(rows));
	scan->children.push_back(move(sink));
	return move(scan);
}

#ifndef GG_REFERENCE_CALLOUTS
//===--------------------------------------------------------------------===//
// The BuildPipelines case
//===--------------------------------------------------------------------===//
// What the reference's own case would do for an operator that is a source over sink children — for every sink the
// steps of its single-child-sink case (executor.cpp:411-413, 478-503), each sink's pipeline depending on the one
// before it, the consuming pipeline on all of them.
//
// libduckdb's BuildPipelines calls itself directly, so only the call from Executor::Initialize passes the shim.
// The rule therefore wraps the whole traversal: it hides the sink children of every graph scan (the reference's
// traversal then takes each scan for the leaf source it is), runs the original, puts the children back and builds
// the sinks' pipelines under the pipelines that ended up reading from a graph scan.
using build_pipelines_fn = void (*)(void *, void *, void *);

//! The graph scans of a plan that is about to be EXECUTED — and, on the way, the tables it is about to write:
//! every execution passes here (Executor::Initialize), also of statements prepared long ago, which the observers
//! of the CreatePlan overloads (gg_plan_rule.cpp, WriteObserver) never see again.
static void CollectGraphScans(PhysicalOperator *op, vector<PhysicalGGGraphScan *> &scans,
                              vector<PhysicalRecursiveCTE *> &ctes) {
	if (!op) {
		return;
	}
	if (auto scan = dynamic_cast<PhysicalGGGraphScan *>(op)) {
		scans.push_back(scan);
		return;
	}
	for (auto &child : op->children) {
		CollectGraphScans(child.get(), scans, ctes);
	}
	switch (op->type) { // the operators whose sub-plans are not their children (executor.cpp:487-495, 522-527)
	case PhysicalOperatorType::EXECUTE:
		CollectGraphScans(((PhysicalExecute &)*op).plan, scans, ctes);
		break;
	case PhysicalOperatorType::DELIM_JOIN:
		CollectGraphScans(((PhysicalDelimJoin &)*op).join.get(), scans, ctes);
		break;
	case PhysicalOperatorType::RECURSIVE_CTE:
		ctes.push_back((PhysicalRecursiveCTE *)op);
		break;
	case PhysicalOperatorType::INSERT:
		if (((PhysicalInsert &)*op).table) {
			GGDropPinsOfTable(((PhysicalInsert &)*op).table->oid);
		}
		break;
	case PhysicalOperatorType::DELETE_OPERATOR:
		GGDropPinsOfTable(((PhysicalDelete &)*op).tableref.oid);
		break;
	case PhysicalOperatorType::UPDATE:
		GGDropPinsOfTable(((PhysicalUpdate &)*op).tableref.oid);
		break;
	default:
		break;
	}
}

static int GGBuildPipelinesRule(void *executor_p, void *op_p, void *current_p) {
	vector<PhysicalGGGraphScan *> scans;
	vector<PhysicalRecursiveCTE *> ctes;
	CollectGraphScans((PhysicalOperator *)op_p, scans, ctes);
	if (scans.empty()) {
		return 0;
	}
	auto original = (void *(*)(int))dlsym(RTLD_DEFAULT, "gg_plan_hook_original");
	auto build = original ? (build_pipelines_fn)original(GG_PLAN_HOOK_PIPELINES) : nullptr;
	if (!build) {
		return 0;
	}
	auto &executor = *(Executor *)executor_p;
	auto current = (Pipeline *)current_p;
	vector<vector<unique_ptr<PhysicalOperator>>> hidden(scans.size());
	for (idx_t i = 0; i < scans.size(); i++) {
		scans[i]->keep_graph = !ctes.empty();
		hidden[i] = move(scans[i]->children);
		scans[i]->children.clear();
	}
	try {
		build(executor_p, op_p, current_p);
	} catch (...) {
		for (idx_t i = 0; i < scans.size(); i++) {
			scans[i]->children = move(hidden[i]);
		}
		throw;
	}
	for (idx_t i = 0; i < scans.size(); i++) {
		scans[i]->children = move(hidden[i]);
	}
	// every pipeline the traversal made (a scan may feed several: union and child pipelines copy their operators)
	vector<Pipeline *> all {current};
	vector<vector<shared_ptr<Pipeline>> *> lists {&executor.pipelines};
	for (auto &entry : executor.union_pipelines) {
		lists.push_back(&entry.second);
	}
	for (auto &entry : executor.child_pipelines) {
		lists.push_back(&entry.second);
	}
	for (auto list : lists) {
		for (auto &pipeline : *list) {
			all.push_back(pipeline.get());
		}
	}
	// the pipelines of a recursive arm are not part of the main schedule: PhysicalRecursiveCTE resets and re-runs them
	// per iteration through Executor::ReschedulePipelines, whose event map holds only those pipelines
	// (executor.cpp:140-170: a dependency on anything else is dereferenced unchecked)
	vector<std::pair<Pipeline *, PhysicalRecursiveCTE *>> inner;
	for (auto cte : ctes) {
		for (auto &pipeline : cte->pipelines) {
			inner.emplace_back(pipeline.get(), cte);
		}
	}
	vector<shared_ptr<Pipeline>> added;
	for (auto scan : scans) {
		vector<Pipeline *> readers;
		PhysicalRecursiveCTE *inside = nullptr;
		for (auto pipeline : all) {
			if (pipeline->source == scan) {
				readers.push_back(pipeline);
			}
		}
		for (auto &entry : inner) {
			if (entry.first->source == scan) {
				readers.push_back(entry.first);
				inside = entry.second;
			}
		}
		// (a prepared plan that is executed again still carries the recursive arm's pipelines of its earlier
		// executions — the reference never clears PhysicalRecursiveCTE::pipelines, executor.cpp:470,503 — so several
		// inner pipelines may name the scan as their source; outside a recursive arm there is exactly one reader)
		if (readers.empty() || (!inside && readers.size() != 1)) {
			throw InternalException(scan->GetName() + ": expected one pipeline to read from the scan");
		}
		if (inside) {
			// A scan in the recursive arm: the graph is invariant over the iterations, so its sinks run ONCE, in the main
			// schedule, before the pipelines that pull from the CTE (the recursion runs inside that source's GetData,
			// physical_recursive_cte.cpp:48-93); the inner reader gets no dependency and finds the graph kept
			readers.clear();
			for (auto pipeline : all) {
				if (pipeline->source == inside) {
					readers.push_back(pipeline);
				}
			}
			if (readers.empty()) {
				throw InternalException(scan->GetName() + ": no pipeline reads from the recursive CTE around the scan");
			}
			scan->keep_graph = true;
		}
		shared_ptr<Pipeline> previous;
		for (auto &child : scan->children) {
			auto pipeline = make_shared<Pipeline>(executor);
			pipeline->sink = child.get();
			child->sink_state.reset();
			if (previous) {
				pipeline->AddDependency(previous);
			}
			for (auto reader : readers) {
				reader->AddDependency(pipeline);
			}
			// (a sink's child is a table scan: its own traversal never meets the recursive CTE)
			build(executor_p, child->children[0].get(), pipeline.get());
			added.push_back(pipeline);
			previous = pipeline;
		}
	}
	for (auto &pipeline : added) {
		executor.pipelines.push_back(pipeline);
	}
	return 1;
}

#endif // GG_REFERENCE_CALLOUTS

static bool g_pipeline_rule = false;

void GGPipelineSinksNative() {
	g_pipeline_rule = true;
}

void GGKeepGraphs(PhysicalOperator &plan) {
	if (auto scan = dynamic_cast<PhysicalGGGraphScan *>(&plan)) {
		scan->keep_graph = true;
		return;
	}
	for (auto &child : plan.children) {
		GGKeepGraphs(*child);
	}
}

void GGRegisterPipelineRule() {
#ifndef GG_REFERENCE_CALLOUTS
	auto reg = (int (*)(int, gg_plan_rule_fn))dlsym(RTLD_DEFAULT, "gg_plan_hook_register");
	auto kinds = (int (*)())dlsym(RTLD_DEFAULT, "gg_plan_hook_kinds");
	if (!reg || !kinds || kinds() <= GG_PLAN_HOOK_PIPELINES || !dlsym(RTLD_DEFAULT, "gg_plan_hook_original")) {
		return; // no shim, or one built before it knew this hook
	}
	g_pipeline_rule = reg(GG_PLAN_HOOK_PIPELINES, GGBuildPipelinesRule) == 0;
#endif
}

} // namespace duckdb

extern "C" int gg_pipeline_rule_registered() {
	return duckdb::g_pipeline_rule ? 1 : 0;
}
