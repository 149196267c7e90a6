// gg_operators.hpp — DuckDB physical operators backed by the MI355X graph kernels (libgg.so).
//
// Host side of the drop-in: these classes implement the reference's own operator interface
// (src/include/duckdb/execution/physical_operator.hpp:24-165 in cwida/duckdb-pgq.old) and call the
// C-ABI of include/gg.h where the reference's operators call their CPU data structures:
//
//   PhysicalGGVertexSink / PhysicalGGEdgeSink   <->  PhysicalHashJoin::Sink/Combine/Finalize
//        (src/execution/operator/join/physical_hash_join.cpp:128-185): the "build side" of the
//        adjacency index; Sink is called concurrently from the pipeline's worker threads with
//        <=1024-row DataChunks, Finalize once.
//   PhysicalGGPathExpand                         <->  the chain of PhysicalHashJoin::Execute probes
//        (physical_hash_join.cpp:217-254) of a k-hop pattern; a source that emits the walks.
//   PhysicalGGShortestPath                       <->  PhysicalRecursiveCTE + min(hop) aggregate
//        (src/execution/operator/set/physical_recursive_cte.cpp:48-139) for the bi-10 friends CTE.
//
// Compiled against the reference's headers; duckdb symbols are resolved by the hosting libduckdb at
// load time (the library is loaded as an extension: gg_duckdb_extension.cpp).
#pragma once

#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <vector>

#include "duckdb.hpp"
#include "duckdb/execution/physical_operator.hpp"
#include "gg.h"

namespace duckdb {
class TableCatalogEntry;
struct GGExpandStream;

//! Device graph shared by the sinks that build it and the sources that query it.
struct GGGraph {
	//! keep_edge_rowids: the CSR keeps the rowid the edge sink gets as its third column (walks with their edges)
	//! parts > 1: the graph is ownership-sharded (gg_csr_build_shard, gg.h) — this object is part 0 and `peers`
	//! are parts 1..parts-1, each with a context of its own on device part % gg_device_count.  The sinks hand every
	//! chunk to every part (a part skips the rows it does not own before the id lookups), the parts build side by
	//! side, and the only source that accepts such a graph is the count of all 2-hop walks (k_max = 2, gg.h), whose per-part
	//! counts add (no data-path collective: SURVEY.md section 8e).
	explicit GGGraph(int device, bool keep_edge_rowids = false, int parts = 1);
	~GGGraph();
	//! number of graphs a plan counting the 2-hop walks of all sources is sharded over: GG_DEVICES (default 1)
	static int ConfiguredParts();
	int Parts() const {
		return 1 + (int)peers.size();
	}
	GGGraph &Part(int p) {
		return p == 0 ? *this : *peers[p - 1];
	}
	//! fn(part index, part) on one thread per part; the first exception is rethrown on the caller's
	void ForEachPart(const std::function<void(int, GGGraph &)> &fn);
	//! Turn a non-zero gg status into the exception the reference's operators would throw.
	static void Check(int rc, const char *what);

	int device = 0;
	gg_ctx *ctx = nullptr;        // taken from / returned to a pool of idle contexts
	gg_csr *csr = nullptr;        // path graph
	gg_csr *filter_csr = nullptr; // optional second edge table over the same vertex set (same-neighbour filter)
	std::mutex lock; // gg calls other than the appends are externally serialised (gg.h)
	vector<unique_ptr<GGGraph>> peers;
	//! Host copy of the path CSR's vertex table (dense index -> id), exported once per build and shared by the
	//! statements that unpack dense indices on the host (a pinned graph serves many: 3.6 MB through pageable memory
	//! per SF100 statement otherwise).  Callers hold `lock`; PhysicalGGEdgeSink::Finalize drops it with the CSR.
	shared_ptr<const vector<int64_t>> VertexIds();
	shared_ptr<const vector<int64_t>> vertex_ids;
};

//! Copy one integer key column of a chunk into `out` (BIGINT or INTEGER physical type, any vector
//! type via Orrify); rows whose key is NULL in ANY of the given columns are skipped, as
//! JoinHashTable::PrepareKeys does for join keys (src/execution/join_hashtable.cpp:126-148).
idx_t GGExtractKeys(DataChunk &input, const vector<idx_t> &cols, vector<vector<int64_t>> &out);
//! Same, without the copy where possible: if every requested column is a flat, all-valid BIGINT vector
//! (what DataTable::Scan produces for NOT NULL BIGINT columns) `keys[c]` points into the chunk itself,
//! otherwise into `scratch` filled by GGExtractKeys.  Returns the number of rows behind each pointer.
idx_t GGKeyColumns(DataChunk &input, const vector<idx_t> &cols, vector<vector<int64_t>> &scratch,
                   vector<const int64_t *> &keys);

//! A thread's window onto a device-resident result table: up to SLAB_ROWS rows of up to GG_MAX_HOPS+1
//! int64 columns in page-locked host memory (gg_host_alloc), refilled with one copy per column and served
//! to the pipeline in <=1024-row DataChunks.  This is the LocalSourceState of the GG sources, so several
//! pipeline threads drain one result concurrently, each through its own slab.
class GGResultSlab : public LocalSourceState {
public:
	//! 2 MB per column and fetch: a result drains at 44 GB/s in copies of that size over the library's fetch lanes
	//! (36 at 1 MB, 48 at 4 MB, 51 at 8 MB — scripts/bench_fetch.py, profiles/r04_bench_fetch.txt; 22-29 GB/s before
	//! the lanes) against twice / four times the page-locked memory per thread
	static constexpr idx_t SLAB_ROWS = 1u << 18;

	explicit GGResultSlab(shared_ptr<GGGraph> graph);
	~GGResultSlab() override;

	//! make room for `columns` columns; returns the column base pointers
	int64_t **Columns(idx_t columns);

	shared_ptr<GGGraph> graph;
	int64_t *memory = nullptr;
	idx_t capacity_columns = 0;
	int64_t *column[GG_MAX_HOPS + 1];
	idx_t rows = 0, pos = 0; // filled rows, next row to serve
	int table = 0;           // which table of the result the slab holds (hop length / batch)
};

class PhysicalGGVertexSink : public PhysicalOperator {
public:
	PhysicalGGVertexSink(shared_ptr<GGGraph> graph, vector<LogicalType> types, idx_t estimated_cardinality);

	shared_ptr<GGGraph> graph;

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
	string GetName() const override {
		return "GG_VERTEX_SINK";
	}
};

//! Input columns: (source key, destination key[, edge rowid]).  Finalize builds the CSR.
class PhysicalGGEdgeSink : public PhysicalOperator {
public:
	//! as_filter: the rows are a SECOND edge table over the already staged vertex set; Finalize builds
	//! GGGraph::filter_csr instead of GGGraph::csr.
	//! derive_vertices: there is no vertex sink — the pattern is a join chain over the edge table alone
	//! (k1.dst = k2.src), so the vertex set is the distinct endpoint ids (gg_vertices_from_edges).
	//! keep_vertices (with derive_vertices): unite the endpoint ids with the vertex ids already staged (a
	//! second edge table over the same id space).  build = false: Finalize stops after the vertex set —
	//! the edge rows only contribute their endpoints (the table is ingested again for its own CSR).
	PhysicalGGEdgeSink(shared_ptr<GGGraph> graph, vector<LogicalType> types, idx_t estimated_cardinality,
	                   bool as_filter = false, bool derive_vertices = false, bool keep_vertices = false,
	                   bool build = true);

	shared_ptr<GGGraph> graph;
	bool as_filter;
	bool derive_vertices;
	bool keep_vertices;
	bool build;
	//! Clears what this sink's options say must leave the staging area before its rows go in; runs once, when the
	//! sink runs (first Sink call or Finalize), never at schedule time.
	void PrepareStaging(GlobalSinkState &gstate) const;

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
	string GetName() const override {
		return "GG_EDGE_SINK";
	}
};

//! Source: walks of length k_min..k_max.  Output: (hops INTEGER, v0 BIGINT, ..., v{k_max} BIGINT),
//! trailing vertices NULL for shorter walks.  count_only: (hops INTEGER, rows BIGINT, digest BIGINT,
//! traversed_edges BIGINT), one row per length.
class PhysicalGGPathExpand : public PhysicalOperator {
public:
	//! rows_only (with count_only): the row counts come from degrees (gg_khop_count) — what `count(*)` over the join
	//! chain needs; the digest column is 0 and no walk is formed.  Without it the counting expansion runs and the
	//! digest of every walk comes with the count (the gg_path_count table function, parity tests, the benchmark).
	PhysicalGGPathExpand(shared_ptr<GGGraph> graph, int k_min, int k_max, bool count_only, vector<int64_t> sources,
	                     bool all_sources, idx_t estimated_cardinality, bool rows_only = false);

	static vector<LogicalType> OutputTypes(int k_max, bool count_only);
	//! expand the current part of a stream of a result that is produced part by part (see GetGlobalSourceState)
	void MaterialisePart(GGExpandStream &stream) const;
	void PlanMiddleParts(GGExpandStream &stream, GGGraph &part, const gg_khop_stats &stats) const;

	shared_ptr<GGGraph> graph;
	int k_min, k_max;
	bool count_only;
	vector<int64_t> sources;
	bool all_sources;
	bool rows_only;

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
		return true;
	}
	string GetName() const override {
		return "GG_PATH_EXPAND";
	}
};

//! Streaming operator + sink: a generic INNER hash join on ONE integer key whose build side is a base-table scan —
//! the reference's PhysicalHashJoin in both of its roles (src/execution/operator/join/physical_hash_join.cpp:128-254):
//!   children[1] (a scan of the build table's key column and rowid) is SUNK into a device index keyed on the key
//!               (Sink / Combine / Finalize = the hash-join build, concurrent Sink calls; gg_edges_append, gg_csr_build);
//!   children[0] (any plan) is PROBED chunk by chunk: Execute sends the chunk's keys to the device (gg_join_probe: for
//!               every probe position the rowids of the build rows under its key), slices the probe chunk by position
//!               and fetches the build table's columns by rowid (DataTable::Fetch, as the index join does) —
//!               HAVE_MORE_OUTPUT while one probe chunk's matches exceed a DataChunk, like ScanStructure::Next.
//! Its PhysicalOperatorType is HASH_JOIN, so the reference's executor schedules it with the code it has for joins
//! (Executor::BuildPipelines, src/parallel/executor.cpp:424-442: the operator is all it looks at), also inside the
//! arms of a recursive CTE.  Output columns: `probe_columns` of the probe child's, then `build_columns` of the build table.
class PhysicalGGKeyJoin : public PhysicalOperator {
public:
	PhysicalGGKeyJoin(vector<LogicalType> types, unique_ptr<PhysicalOperator> probe, unique_ptr<PhysicalOperator> build_scan,
	                  idx_t probe_key, vector<idx_t> probe_columns, TableCatalogEntry *build_table, column_t build_key,
	                  vector<column_t> build_columns, idx_t estimated_cardinality);

	idx_t probe_key;                 // column of the probe child's chunks that holds the key
	vector<idx_t> probe_columns;     // columns of the probe child's chunks that go out, in output order
	TableCatalogEntry *build_table;
	column_t build_key;
	vector<column_t> build_columns;  // of the build table, in output order (COLUMN_IDENTIFIER_ROW_ID allowed)

public:
	// sink (build side)
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
	// operator (probe side)
	unique_ptr<OperatorState> GetOperatorState(ClientContext &context) const override;
	OperatorResultType Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
	                           OperatorState &state) const override;
	bool ParallelOperator() const override {
		return true;
	}
	bool RequiresCache() const override {
		return true;
	}
	string GetName() const override {
		return "GG_KEY_JOIN";
	}
	string ParamsToString() const override;
};

//! Source: `hops`-hop walks WITH payload columns of the edge table (a join chain that projects columns of its edge
//! instances other than the two keys).  The device returns every walk with the rowid of each edge taken
//! (gg_expand_khop_edges); the payload columns are then fetched from the base table by rowid, in the statement's
//! transaction — the late-materialising counterpart of the gather the reference's hash join does per match from its
//! build side (src/execution/join_hashtable.cpp:466-473), through the call its index join uses
//! (DataTable::Fetch, src/execution/operator/join/physical_index_join.cpp).
//! Output: (hops INTEGER, v0 BIGINT, ..., v{hops} BIGINT, payload columns in the order given).
class PhysicalGGPathEdges : public PhysicalOperator {
public:
	//! payload: (1-based edge number of the walk, column of the edge table)
	PhysicalGGPathEdges(shared_ptr<GGGraph> graph, int hops, vector<int64_t> sources, bool all_sources,
	                    TableCatalogEntry *edge_table, vector<std::pair<idx_t, column_t>> payload,
	                    idx_t estimated_cardinality);
	static vector<LogicalType> OutputTypes(int hops, TableCatalogEntry &edge_table,
	                                       const vector<std::pair<idx_t, column_t>> &payload);

	shared_ptr<GGGraph> graph;
	int hops;
	vector<int64_t> sources;
	bool all_sources;
	TableCatalogEntry *edge_table;
	vector<std::pair<idx_t, column_t>> payload;

	//! expand the current part (slices of the sources under GG_RESULT_BUDGET_MB; caller holds graph->lock)
	void MaterialisePart(GlobalSourceState &gstate) const;

public:
	unique_ptr<GlobalSourceState> GetGlobalSourceState(ClientContext &context) const override;
	void GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate,
	             LocalSourceState &lstate) const override;
	bool IsSource() const override {
		return true;
	}
	string GetName() const override {
		return "GG_PATH_EDGES";
	}
};

//! Source: fixed-length walks whose vertices all share a neighbour in the filter graph — Train Benchmark
//! ConnectedSegments (benchmark/trainbenchmark/queries/connectedsegments.sql:1-25): `hops` connectsTo
//! edges from the source segments, all hops+1 segments monitored by one sensor.
//! Output: (w BIGINT, v0 BIGINT, ..., v{hops} BIGINT).
class PhysicalGGFilteredPaths : public PhysicalOperator {
public:
	PhysicalGGFilteredPaths(shared_ptr<GGGraph> graph, int hops, vector<int64_t> sources,
	                        idx_t estimated_cardinality, bool all_sources = false);

	shared_ptr<GGGraph> graph;
	int hops;
	vector<int64_t> sources;
	bool all_sources; // walks may start at any vertex (no source table in the pattern)

public:
	unique_ptr<GlobalSourceState> GetGlobalSourceState(ClientContext &context) const override;
	void GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate,
	             LocalSourceState &lstate) const override;
	bool IsSource() const override {
		return true;
	}
	string GetName() const override {
		return "GG_FILTERED_PATHS";
	}
};

//! Source: (vertex id BIGINT, h1 BIGINT, ..., hk BIGINT) — one row per vertex that ends a walk of 1..k_max edges from
//! one of the sources; h = 1 iff a walk of exactly h edges ends there (gg_walk_endpoints).  The device form of the
//! reference's hash-aggregate dedupe above a UNION of endpoint sets (interactive-complex-3.sql:3-12).
class PhysicalGGWalkEndpoints : public PhysicalOperator {
public:
	PhysicalGGWalkEndpoints(shared_ptr<GGGraph> graph, vector<int64_t> sources, int k_max, idx_t estimated_cardinality);

	static vector<LogicalType> OutputTypes(int k_max);

	shared_ptr<GGGraph> graph;
	vector<int64_t> sources;
	int k_max;

public:
	unique_ptr<GlobalSourceState> GetGlobalSourceState(ClientContext &context) const override;
	void GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate,
	             LocalSourceState &lstate) const override;
	bool IsSource() const override {
		return true;
	}
	string GetName() const override {
		return "GG_WALK_ENDPOINTS";
	}
};

//! Source: (startPerson BIGINT, friend BIGINT, hopCount INTEGER) for every pair reached within
//! max_hops — the friends_shortest relation of benchmark/ldbc/queries/bi-10-shortestpath.sql:8-31.
//! Sources are processed in batches of 64 bit lanes.
class PhysicalGGShortestPath : public PhysicalOperator {
public:
	//! lone_sources: a source that is not a vertex of the graph (it has no edge in an edge-only vertex
	//! set) still reaches itself at hop 0 — the seed row of the recursive CTE does not depend on the edges
	PhysicalGGShortestPath(shared_ptr<GGGraph> graph, vector<int64_t> sources, int max_hops,
	                       idx_t estimated_cardinality, bool lone_sources = false);

	//! BFS of the next 64-source batch (batches run one at a time, when the previous one is drained)
	void RunBatch(GlobalSourceState &gstate) const;

	shared_ptr<GGGraph> graph;
	vector<int64_t> sources;
	int max_hops;
	bool lone_sources;

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
		return true;
	}
	string GetName() const override {
		return "GG_SHORTEST_PATH";
	}
};

} // namespace duckdb
