// gg_pipeline.hpp — the device graph's build side as pipeline sinks of the reference's executor (gg_pipeline.cpp)
#pragma once

#include <functional>

#include "gg_extension.hpp"
#include "gg_operators.hpp"

namespace duckdb {

//! Where the sinks of one plan leave the graph they build and its scan picks it up; one graph per execution.
struct GGGraphSlot {
	mutex lock;
	shared_ptr<GGGraph> graph;
	int shards = 1; // GGGraphSpec::shards of the plan
};

//! PhysicalGGVertexSink / PhysicalGGEdgeSink created when the pipeline starts (the device context with them), so a
//! plan that is only explained or prepared never touches the GPU.
class PhysicalGGLazySink : public PhysicalOperator {
public:
	enum Kind { VERTICES, EDGES, EDGES_DERIVE_VERTICES, EDGES_CUSTOM };
	PhysicalGGLazySink(shared_ptr<GGGraphSlot> slot, Kind kind, vector<LogicalType> types, idx_t estimated_cardinality);
	//! an edge sink with PhysicalGGEdgeSink's own switches (plans over two edge tables: ConnectedSegments).  `first`: the
	//! sink that opens the execution's graph; `clear_edges_after`: its Finalize drops the staged edge rows once its own
	//! CSR is built (gg_staging_clear_edges: the vertex numbering stays) — every sink's global state is created when
	//! the pipelines are scheduled, before any of them runs, so nothing of that kind can happen there
	struct EdgeOptions {
		bool first = false, clear_edges_after = false;
		bool as_filter = false, derive_vertices = false, keep_vertices = false, build = true;
	};
	PhysicalGGLazySink(shared_ptr<GGGraphSlot> slot, EdgeOptions options, vector<LogicalType> types,
	                   idx_t estimated_cardinality);

	shared_ptr<GGGraphSlot> slot;
	Kind kind;
	EdgeOptions options;
	mutable unique_ptr<PhysicalOperator> inner;

public:
	unique_ptr<GlobalSinkState> GetGlobalSinkState(ClientContext &context) const override;
	unique_ptr<LocalSinkState> GetLocalSinkState(ExecutionContext &context) const override;
	SinkResultType Sink(ExecutionContext &context, GlobalSinkState &gstate, LocalSinkState &lstate,
	                    DataChunk &input) const override;
	void Combine(ExecutionContext &context, GlobalSinkState &gstate, LocalSinkState &lstate) const override;
	SinkFinalizeType Finalize(Pipeline &pipeline, Event &event, ClientContext &context,
	                          GlobalSinkState &gstate) const override;
	bool IsSink() const override {
		return true;
	}
	bool ParallelSink() const override {
		return true;
	}
	string GetName() const override;
};

//! A GG source over the graph its sink children build.  `factory` makes the actual source operator
//! (PhysicalGGPathExpand, PhysicalGGWalkEndpoints, ...) once the graph exists.
class PhysicalGGGraphScan : public PhysicalOperator {
public:
	//! (the context is the executing statement's: a factory may read more rows in its transaction, e.g. BFS seeds)
	using Factory = std::function<unique_ptr<PhysicalOperator>(ClientContext &, shared_ptr<GGGraph>)>;
	PhysicalGGGraphScan(vector<LogicalType> types, string name, string description, shared_ptr<GGGraphSlot> slot,
	                    Factory factory, bool parallel_result, idx_t estimated_cardinality);

	string name, description;
	shared_ptr<GGGraphSlot> slot;
	Factory factory;
	bool parallel_result;
	//! set while the plan's pipelines are built: the scan sits in a plan with a recursive CTE, whose pipelines are
	//! reset and re-run per iteration — the graph then lives as long as the plan
	mutable bool keep_graph = false;

public:
	unique_ptr<GlobalSourceState> GetGlobalSourceState(ClientContext &context) const override;
	unique_ptr<LocalSourceState> GetLocalSourceState(ExecutionContext &context,
	                                                 GlobalSourceState &gstate) const override;
	void GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate,
	             LocalSourceState &lstate) const override;
	bool IsSource() const override {
		return true;
	}
	bool ParallelSource() const override {
		return parallel_result;
	}
	string GetName() const override;
	string ParamsToString() const override;
};

//! true if a plan over `spec` can read its tables through pipeline sinks: the BuildPipelines rule is registered with
//! the shim, every source is a plain table, and the connection did not ask for pinned graphs
bool GGPipelineSinksAvailable(ClientContext &context, const GGGraphSpec &spec);
//! GG_<name> scan with the sinks (and the reference's own table scans) of `spec` as children
unique_ptr<PhysicalOperator> GGMakeGraphScan(const GGGraphSpec &spec, vector<LogicalType> types, string name,
                                             string description, bool parallel_result,
                                             PhysicalGGGraphScan::Factory factory, idx_t estimated_cardinality);
//! the same for a plan whose graph takes several edge-table passes: one sink child per entry, in build order
struct GGSinkSpec {
	GGScanSource rows;
	PhysicalGGLazySink::EdgeOptions options;
};
unique_ptr<PhysicalOperator> GGMakeGraphScan(const vector<GGSinkSpec> &sinks, vector<LogicalType> types, string name,
                                             string description, bool parallel_result,
                                             PhysicalGGGraphScan::Factory factory, idx_t estimated_cardinality);
//! PhysicalTableScan of the given columns of a base table (what plan_get.cpp:47-60 builds for a seq_scan)
unique_ptr<PhysicalOperator> GGBaseTableScan(const GGScanSource &source);
void GGRegisterPipelineRule();
//! the hosting reference has the BuildPipelines case itself (oracle/callout.patch): pipeline sinks without a rule
void GGPipelineSinksNative();
//! every graph scan below `plan` keeps its graph for as long as the plan lives (plans with a recursive CTE)
void GGKeepGraphs(PhysicalOperator &plan);

} // namespace duckdb

extern "C" int gg_pipeline_rule_registered();
