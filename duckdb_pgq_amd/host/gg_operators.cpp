// gg_operators.cpp — see gg_operators.hpp.
#include "gg_operators.hpp"

#include <cstdlib>
#include <thread>

#include "duckdb/common/exception.hpp"
#include "duckdb/common/types/data_chunk.hpp"
#include "duckdb/common/types/vector.hpp"
#include "duckdb/main/client_context.hpp"
#include "duckdb/catalog/catalog_entry/table_catalog_entry.hpp"
#include "duckdb/storage/data_table.hpp"
#include "duckdb/transaction/transaction.hpp"

namespace duckdb {

//! A slab fetch in flight: the counter was raised under the state's lock when the range was claimed; it comes
//! down when the fetch is over — also when it ends in an exception, or the next claimer would wait for ever.
struct FetchClaim {
	std::atomic<idx_t> &counter;
	FetchClaim(std::atomic<idx_t> &counter_p, bool already_counted) : counter(counter_p) {
		if (!already_counted) {
			counter++;
		}
	}
	~FetchClaim() {
		counter--;
	}
};

//===--------------------------------------------------------------------===//
// GGGraph
//===--------------------------------------------------------------------===//
shared_ptr<const vector<int64_t>> GGGraph::VertexIds() {
	if (!vertex_ids) {
		uint64_t V = 0;
		Check(gg_csr_info(csr, &V, nullptr, nullptr), "gg_csr_info");
		auto ids = make_shared<vector<int64_t>>(V);
		Check(gg_csr_export(csr, nullptr, nullptr, nullptr, ids->data()), "gg_csr_export");
		vertex_ids = move(ids);
	}
	return vertex_ids;
}

void GGGraph::Check(int rc, const char *what) {
	if (rc != GG_OK) {
		throw IOException(string(what) + ": " + gg_last_error());
	}
}

//! Idle device contexts.  Creating one costs tens of milliseconds (pinned staging blocks, stream) and its
//! block cache makes the next build allocation-free, so queries hand them back instead of destroying them.
struct GGContextPool {
	std::mutex lock;
	vector<std::pair<int, gg_ctx *>> idle;
	// (idle contexts are deliberately not destroyed at process exit: the HIP runtime may be gone by then)
	gg_ctx *Acquire(int device) {
		lock_guard<mutex> guard(lock);
		for (idx_t i = 0; i < idle.size(); i++) {
			if (idle[i].first == device) {
				auto ctx = idle[i].second;
				idle.erase(idle.begin() + i);
				return ctx;
			}
		}
		return nullptr;
	}
	void Release(int device, gg_ctx *ctx) {
		lock_guard<mutex> guard(lock);
		if (idle.size() < 4) {
			idle.emplace_back(device, ctx);
		} else {
			gg_ctx_destroy(ctx);
		}
	}
};
static GGContextPool g_context_pool;

int GGGraph::ConfiguredParts() {
	auto env = std::getenv("GG_DEVICES");
	const long n = env ? std::strtol(env, nullptr, 10) : 1;
	return (int)MaxValue<long>(1, MinValue<long>(n, 64));
}

void GGGraph::ForEachPart(const std::function<void(int, GGGraph &)> &fn) {
	if (peers.empty()) {
		fn(0, *this);
		return;
	}
	// one host thread per part: each part's calls go to its own context (own device, own stream), so the
	// parts' uploads and kernels run side by side
	vector<std::thread> threads;
	vector<std::exception_ptr> errors(Parts());
	for (int p = 0; p < Parts(); p++) {
		threads.emplace_back([&, p] {
			try {
				fn(p, Part(p));
			} catch (...) {
				errors[p] = std::current_exception();
			}
		});
	}
	for (auto &thread : threads) {
		thread.join();
	}
	for (auto &error : errors) {
		if (error) {
			std::rethrow_exception(error);
		}
	}
}

GGGraph::GGGraph(int device_p, bool keep_edge_rowids, int parts) : device(device_p) {
	if (parts > 1) {
		int devices = 0;
		Check(gg_device_count(&devices), "gg_device_count");
		for (int p = 1; p < parts; p++) {
			peers.push_back(make_unique<GGGraph>((device_p + p) % MaxValue(devices, 1), keep_edge_rowids, 1));
		}
	}
	ctx = g_context_pool.Acquire(device);
	if (ctx) {
		Check(gg_staging_clear(ctx), "gg_staging_clear");
	} else {
		Check(gg_ctx_create(device, &ctx), "gg_ctx_create");
	}
	// only walks with payload columns of their edges return edge ids: every other plan builds without the
	// edge-rowid payload, as the reference's hash-join build side carries only the columns the query references
	Check(gg_ctx_set_edge_rowid(ctx, keep_edge_rowids ? 1 : 0), "gg_ctx_set_edge_rowid");
}

GGGraph::~GGGraph() {
	if (filter_csr) {
		gg_csr_destroy(filter_csr);
	}
	if (csr) {
		gg_csr_destroy(csr);
	}
	if (ctx) {
		g_context_pool.Release(device, ctx);
	}
}

template <class T>
static void CopyColumn(VectorData &vdata, idx_t count, const vector<bool> &keep, vector<int64_t> &out) {
	auto data = (const T *)vdata.data;
	for (idx_t i = 0; i < count; i++) {
		if (keep[i]) {
			out.push_back((int64_t)data[vdata.sel->get_index(i)]);
		}
	}
}

idx_t GGExtractKeys(DataChunk &input, const vector<idx_t> &cols, vector<vector<int64_t>> &out) {
	const idx_t count = input.size();
	vector<VectorData> vdata(cols.size());
	vector<bool> keep(count, true);
	for (idx_t c = 0; c < cols.size(); c++) {
		// inputs may be flat, constant, dictionary or sequence vectors (e.g. rowid): Orrify gives a
		// uniform (selection, data, validity) view  (src/include/duckdb/common/types/vector.hpp:119)
		input.data[cols[c]].Orrify(count, vdata[c]);
		if (!vdata[c].validity.AllValid()) {
			for (idx_t i = 0; i < count; i++) {
				if (!vdata[c].validity.RowIsValid(vdata[c].sel->get_index(i))) {
					keep[i] = false;
				}
			}
		}
	}
	out.resize(cols.size());
	idx_t kept = 0;
	for (idx_t i = 0; i < count; i++) {
		kept += keep[i];
	}
	for (idx_t c = 0; c < cols.size(); c++) {
		out[c].clear();
		out[c].reserve(kept);
		switch (input.data[cols[c]].GetType().InternalType()) {
		case PhysicalType::INT64:
			CopyColumn<int64_t>(vdata[c], count, keep, out[c]);
			break;
		case PhysicalType::INT32:
			CopyColumn<int32_t>(vdata[c], count, keep, out[c]);
			break;
		case PhysicalType::UINT32:
			CopyColumn<uint32_t>(vdata[c], count, keep, out[c]);
			break;
		case PhysicalType::INT16:
			CopyColumn<int16_t>(vdata[c], count, keep, out[c]);
			break;
		default:
			throw NotImplementedException("GG graph operators need integer key columns");
		}
	}
	return kept;
}

idx_t GGKeyColumns(DataChunk &input, const vector<idx_t> &cols, vector<vector<int64_t>> &scratch,
                   vector<const int64_t *> &keys) {
	keys.resize(cols.size());
	bool direct = true;
	for (idx_t c = 0; c < cols.size() && direct; c++) {
		auto &vec = input.data[cols[c]];
		direct = vec.GetVectorType() == VectorType::FLAT_VECTOR && vec.GetType().InternalType() == PhysicalType::INT64 &&
		         FlatVector::Validity(vec).AllValid();
	}
	if (direct) {
		for (idx_t c = 0; c < cols.size(); c++) {
			keys[c] = FlatVector::GetData<int64_t>(input.data[cols[c]]);
		}
		return input.size();
	}
	const idx_t kept = GGExtractKeys(input, cols, scratch);
	for (idx_t c = 0; c < cols.size(); c++) {
		keys[c] = scratch[c].data();
	}
	return kept;
}

//===--------------------------------------------------------------------===//
// Result slab
//===--------------------------------------------------------------------===//
GGResultSlab::GGResultSlab(shared_ptr<GGGraph> graph_p) : graph(move(graph_p)) {
}

GGResultSlab::~GGResultSlab() {
	if (memory) {
		gg_host_free(graph->ctx, memory);
	}
}

int64_t **GGResultSlab::Columns(idx_t columns) {
	if (columns > capacity_columns) {
		if (memory) {
			gg_host_free(graph->ctx, memory);
			memory = nullptr;
		}
		void *p = nullptr;
		GGGraph::Check(gg_host_alloc(graph->ctx, columns * SLAB_ROWS * sizeof(int64_t), &p), "gg_host_alloc");
		memory = (int64_t *)p;
		capacity_columns = columns;
	}
	for (idx_t c = 0; c < columns; c++) {
		column[c] = memory + c * SLAB_ROWS;
	}
	return column;
}

//===--------------------------------------------------------------------===//
// Sinks
//===--------------------------------------------------------------------===//
class GGSinkGlobalState : public GlobalSinkState {
public:
	std::atomic<idx_t> rows {0};
	//! What a sink does to the staging area before its own rows go in (PhysicalGGEdgeSink::PrepareStaging) happens when
	//! the sink RUNS — its first Sink call, or its Finalize when the table is empty — not when its state is made:
	//! Pipeline::Ready makes every sink's global state at schedule time (pipeline.cpp:125-142), before the sinks ahead of
	//! it in the build order have staged anything.
	std::atomic<bool> prepared {false};
	std::mutex prepare_lock;
};

//! Batch buffers of the edge sinks, kept across statements: a statement's 256 Sink threads would otherwise each
//! allocate, zero and page-fault a buffer they fill two or three times (SF100: 40 M rows over 256 threads).
struct GGBatchPool {
	static constexpr idx_t MAX_IDLE = 256; // (384 KB each at the default batch size: at most 96 MB kept)
	std::mutex lock;
	vector<std::pair<idx_t, int64_t *>> idle; // (capacity in int64 values, memory)
	int64_t *Acquire(idx_t values) {
		{
			lock_guard<mutex> guard(lock);
			for (idx_t i = 0; i < idle.size(); i++) {
				if (idle[i].first == values) {
					auto memory = idle[i].second;
					idle.erase(idle.begin() + i);
					return memory;
				}
			}
		}
		return new int64_t[values]; // (not value-initialised: every row is written before it is read)
	}
	void Release(idx_t values, int64_t *memory) {
		{
			lock_guard<mutex> guard(lock);
			if (idle.size() < MAX_IDLE) {
				idle.emplace_back(values, memory);
				return;
			}
		}
		delete[] memory;
	}
};
static GGBatchPool g_batch_pool;

class GGSinkLocalState : public LocalSinkState {
public:
	~GGSinkLocalState() override {
		if (batch_memory) {
			g_batch_pool.Release(3 * batch_capacity, batch_memory);
		}
	}
	vector<vector<int64_t>> columns; // per-thread conversion buffers, reused across chunks
	vector<const int64_t *> keys;    // what is handed to gg_*_append: the chunk's own vectors or `columns`
	// Edge rows are batched per thread and handed to the staging area BATCH_ROWS at a time: one reservation
	// per 16 chunks instead of one per chunk keeps the Sink threads off each other's reservation lock
	// (GG_SINK_BATCH_ROWS; 1024 = unbatched).  Three columns of batch_capacity rows, from g_batch_pool.
	int64_t *batch_memory = nullptr;
	idx_t batch_capacity = 0;
	int64_t *batch[3] = {nullptr, nullptr, nullptr};
	bool batch_has_rowid = false;
	idx_t batch_rows = 0;
};

static idx_t SinkBatchRows() {
	static const idx_t rows = [] {
		auto env = std::getenv("GG_SINK_BATCH_ROWS");
		const idx_t n = env ? (idx_t)std::strtoull(env, nullptr, 10) : 16 * STANDARD_VECTOR_SIZE;
		return MaxValue<idx_t>(n, STANDARD_VECTOR_SIZE);
	}();
	return rows;
}

static void FlushEdgeBatch(GGGraph &graph, GGSinkLocalState &lstate, bool has_rowid) {
	if (lstate.batch_rows == 0) {
		return;
	}
	for (int p = 0; p < graph.Parts(); p++) {
		GGGraph::Check(gg_edges_append(graph.Part(p).ctx, lstate.batch[0], lstate.batch[1],
		                               has_rowid ? lstate.batch[2] : nullptr, lstate.batch_rows),
		               "gg_edges_append");
	}
	lstate.batch_rows = 0;
}

PhysicalGGVertexSink::PhysicalGGVertexSink(shared_ptr<GGGraph> graph_p, vector<LogicalType> types,
                                           idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::INVALID, move(types), estimated_cardinality), graph(move(graph_p)) {
}

unique_ptr<GlobalSinkState> PhysicalGGVertexSink::GetGlobalSinkState(ClientContext &context) const {
	for (int p = 0; p < graph->Parts(); p++) {
		GGGraph::Check(gg_staging_clear(graph->Part(p).ctx), "gg_staging_clear");
	}
	return make_unique<GGSinkGlobalState>();
}

unique_ptr<LocalSinkState> PhysicalGGVertexSink::GetLocalSinkState(ExecutionContext &context) const {
	return make_unique<GGSinkLocalState>();
}

SinkResultType PhysicalGGVertexSink::Sink(ExecutionContext &context, GlobalSinkState &gstate_p,
                                          LocalSinkState &lstate_p, DataChunk &input) const {
	auto &gstate = (GGSinkGlobalState &)gstate_p;
	auto &lstate = (GGSinkLocalState &)lstate_p;
	idx_t n = GGKeyColumns(input, {0}, lstate.columns, lstate.keys);
	// thread-safe append (gg.h): one call per DataChunk, like JoinHashTable::Build per Sink call
	for (int p = 0; p < graph->Parts(); p++) {
		GGGraph::Check(gg_vertices_append(graph->Part(p).ctx, lstate.keys[0], n), "gg_vertices_append");
	}
	gstate.rows += n;
	return SinkResultType::NEED_MORE_INPUT;
}

void PhysicalGGVertexSink::Combine(ExecutionContext &context, GlobalSinkState &gstate, LocalSinkState &lstate) const {
}

SinkFinalizeType PhysicalGGVertexSink::Finalize(Pipeline &pipeline, Event &event, ClientContext &context,
                                                GlobalSinkState &gstate) const {
	return SinkFinalizeType::READY;
}

PhysicalGGEdgeSink::PhysicalGGEdgeSink(shared_ptr<GGGraph> graph_p, vector<LogicalType> types,
                                       idx_t estimated_cardinality, bool as_filter_p, bool derive_vertices_p,
                                       bool keep_vertices_p, bool build_p)
    : PhysicalOperator(PhysicalOperatorType::INVALID, move(types), estimated_cardinality), graph(move(graph_p)),
      as_filter(as_filter_p), derive_vertices(derive_vertices_p), keep_vertices(keep_vertices_p), build(build_p) {
}

unique_ptr<GlobalSinkState> PhysicalGGEdgeSink::GetGlobalSinkState(ClientContext &context) const {
	if (graph->Parts() > 1 && (as_filter || !build)) {
		throw InternalException("a sharded graph has one edge table");
	}
	return make_unique<GGSinkGlobalState>();
}

void PhysicalGGEdgeSink::PrepareStaging(GlobalSinkState &gstate_p) const {
	auto &gstate = (GGSinkGlobalState &)gstate_p;
	if (gstate.prepared.load(std::memory_order_acquire)) {
		return;
	}
	lock_guard<mutex> guard(gstate.prepare_lock);
	if (gstate.prepared.load(std::memory_order_relaxed)) {
		return;
	}
	if (as_filter) {
		// second edge table over the same staged vertices: drop the first table's staged rows only
		GGGraph::Check(gg_staging_clear_edges(graph->ctx), "gg_staging_clear_edges");
	} else if (derive_vertices && !keep_vertices) {
		// no vertex sink ran before this one: start from empty staging
		for (int p = 0; p < graph->Parts(); p++) {
			GGGraph::Check(gg_staging_clear(graph->Part(p).ctx), "gg_staging_clear");
		}
	}
	gstate.prepared.store(true, std::memory_order_release);
}

unique_ptr<LocalSinkState> PhysicalGGEdgeSink::GetLocalSinkState(ExecutionContext &context) const {
	return make_unique<GGSinkLocalState>();
}

SinkResultType PhysicalGGEdgeSink::Sink(ExecutionContext &context, GlobalSinkState &gstate_p, LocalSinkState &lstate_p,
                                        DataChunk &input) const {
	auto &gstate = (GGSinkGlobalState &)gstate_p;
	auto &lstate = (GGSinkLocalState &)lstate_p;
	PrepareStaging(gstate_p);
	const bool has_rowid = input.ColumnCount() >= 3;
	idx_t n = has_rowid ? GGKeyColumns(input, {0, 1, 2}, lstate.columns, lstate.keys)
	                    : GGKeyColumns(input, {0, 1}, lstate.columns, lstate.keys);
	const idx_t capacity = SinkBatchRows();
	const idx_t ncols = has_rowid ? 3 : 2;
	if (!lstate.batch_memory) {
		lstate.batch_memory = g_batch_pool.Acquire(3 * capacity);
		lstate.batch_capacity = capacity;
		for (idx_t c = 0; c < 3; c++) {
			lstate.batch[c] = lstate.batch_memory + c * capacity;
		}
	}
	lstate.batch_has_rowid = has_rowid;
	if (lstate.batch_rows + n > capacity) {
		FlushEdgeBatch(*graph, lstate, has_rowid);
	}
	for (idx_t c = 0; c < ncols; c++) {
		memcpy(lstate.batch[c] + lstate.batch_rows, lstate.keys[c], n * sizeof(int64_t));
	}
	lstate.batch_rows += n;
	gstate.rows += n;
	return SinkResultType::NEED_MORE_INPUT;
}

void PhysicalGGEdgeSink::Combine(ExecutionContext &context, GlobalSinkState &gstate, LocalSinkState &lstate_p) const {
	auto &lstate = (GGSinkLocalState &)lstate_p;
	FlushEdgeBatch(*graph, lstate, lstate.batch_has_rowid);
}

SinkFinalizeType PhysicalGGEdgeSink::Finalize(Pipeline &pipeline, Event &event, ClientContext &context,
                                              GlobalSinkState &gstate) const {
	// single-threaded, after every Sink/Combine (physical_operator.hpp:145-147): build the index
	PrepareStaging(gstate); // (an empty table never reached Sink)
	lock_guard<mutex> guard(graph->lock);
	if (graph->Parts() > 1) {
		// every part saw every row; each derives the same vertex numbering (sorted distinct endpoint ids) or staged
		// the same vertex table, and keeps the CSR rows of the vertices it owns
		const int parts = graph->Parts();
		graph->ForEachPart([&](int p, GGGraph &part) {
			part.vertex_ids.reset();
			if (part.csr) {
				gg_csr_destroy(part.csr);
				part.csr = nullptr;
			}
			if (derive_vertices) {
				GGGraph::Check(gg_vertices_from_edges(part.ctx, keep_vertices ? 1 : 0, nullptr),
				               "gg_vertices_from_edges");
			}
			GGGraph::Check(gg_csr_build_shard(part.ctx, p, parts, &part.csr), "gg_csr_build_shard");
		});
		return SinkFinalizeType::READY;
	}
	gg_csr *&target = as_filter ? graph->filter_csr : graph->csr;
	if (!as_filter) {
		graph->vertex_ids.reset(); // (of the CSR that goes)
	}
	if (target) {
		gg_csr_destroy(target);
		target = nullptr;
	}
	if (derive_vertices) {
		GGGraph::Check(gg_vertices_from_edges(graph->ctx, keep_vertices ? 1 : 0, nullptr), "gg_vertices_from_edges");
	}
	if (!build) {
		return SinkFinalizeType::READY;
	}
	GGGraph::Check(gg_csr_build(graph->ctx, &target), "gg_csr_build");
	return SinkFinalizeType::READY;
}

//===--------------------------------------------------------------------===//
// Path expansion source
//===--------------------------------------------------------------------===//
//! One producer of a materialised result: the graph part (device context) whose walks it makes, the ranges still to
//! produce, and the part that stands in HBM right now with its scan position.  An unsharded graph has one stream; an
//! ownership-sharded one (GG_DEVICES) one per part — the pipeline's threads drain them side by side, each stream's
//! rows crossing its own device's PCIe link.
struct GGExpandStream {
	~GGExpandStream() {
		if (result) {
			gg_result_destroy(result);
		}
	}
	int graph_part = 0;
	bool by_middle = false;        // ranges are middle-vertex ranges (gg_expand_khop_mid_result), else source ranges / slices
	gg_khop_stats stats;           // of the part that is materialised right now
	gg_result *result = nullptr;   // walks of the current part, in HBM
	// A result too large for the device-memory budget is produced part by part, one materialised at a time.
	// parts[i] = [first, last): middle vertices, source vertices (all sources) or positions of the source list.
	vector<std::pair<uint64_t, uint64_t>> parts;
	idx_t part = 0;
	bool started = false, done = false;
	std::atomic<idx_t> fetching {0}; // slab fetches still reading `result` (it may not be freed under them)
	// scan position: (current hop length, next row of it nobody has claimed); pipeline threads claim
	// GGResultSlab::SLAB_ROWS rows at a time under the lock and fetch them into their own slab
	mutex lock;
	int hop = 0;
	idx_t offset = 0;
};

class GGExpandGlobalState : public GlobalSourceState {
public:
	idx_t MaxThreads() override {
		return max_threads;
	}

	gg_khop_stats stats; // count_only: of everything
	mutex lock;          // count_only: the one row set is handed out once
	int hop = 0;
	vector<unique_ptr<GGExpandStream>> streams;
	std::atomic<idx_t> next_stream {0}; // threads start on different streams
	idx_t max_threads = 1;
};

//! The slab of a thread plus the stream it drains at the moment.
class GGExpandLocalState : public GGResultSlab {
public:
	explicit GGExpandLocalState(shared_ptr<GGGraph> graph) : GGResultSlab(move(graph)) {
	}
	idx_t stream = INVALID_INDEX;
};

vector<LogicalType> PhysicalGGPathExpand::OutputTypes(int k_max, bool count_only) {
	vector<LogicalType> types;
	types.push_back(LogicalType::INTEGER);
	if (count_only) {
		types.push_back(LogicalType::BIGINT);
		types.push_back(LogicalType::BIGINT);
		types.push_back(LogicalType::BIGINT);
	} else {
		for (int c = 0; c <= k_max; c++) {
			types.push_back(LogicalType::BIGINT);
		}
	}
	return types;
}

PhysicalGGPathExpand::PhysicalGGPathExpand(shared_ptr<GGGraph> graph_p, int k_min_p, int k_max_p, bool count_only_p,
                                           vector<int64_t> sources_p, bool all_sources_p,
                                           idx_t estimated_cardinality, bool rows_only_p)
    : PhysicalOperator(PhysicalOperatorType::INVALID, OutputTypes(k_max_p, count_only_p), estimated_cardinality),
      graph(move(graph_p)), k_min(k_min_p), k_max(k_max_p), count_only(count_only_p), sources(move(sources_p)),
      all_sources(all_sources_p), rows_only(rows_only_p && count_only_p) {
}

//! A source list for the C-ABI: an EMPTY list must not arrive as a null pointer (gg.h: null means every vertex).
static const int64_t *SourceIds(const vector<int64_t> &sources) {
	static const int64_t none = 0;
	return sources.empty() ? &none : sources.data();
}

//! Row counts per walk length from degrees (gg_khop_count) in the shape the counting expansion reports them:
//! digests 0, traversed edges = the rows of every length up to k_max (one adjacency entry per row).
static void CountRowsFromDegrees(gg_ctx *ctx, const gg_csr *csr, const vector<int64_t> &sources, bool all_sources,
                                 int k_min, int k_max, gg_khop_stats &stats) {
	memset(&stats, 0, sizeof(stats));
	uint64_t rows[GG_MAX_HOPS + 1];
	GGGraph::Check(gg_khop_count(ctx, csr, all_sources ? nullptr : SourceIds(sources), sources.size(), 1, k_max, rows),
	               "gg_khop_count");
	for (int h = 1; h <= k_max; h++) {
		stats.traversed_edges += rows[h];
		if (h >= k_min) {
			stats.rows[h] = rows[h];
		}
	}
}

//! Device-memory budget for one materialised part (GG_RESULT_BUDGET_MB; default 40 GiB of the 288: three id columns
//! of ~13 GiB each — parts whose columns are below ~8 GiB are written slower, csrc/gg_runtime.hip "Placement").
static uint64_t ResultBudgetBytes() {
	auto env = std::getenv("GG_RESULT_BUDGET_MB");
	const uint64_t mb = env ? std::strtoull(env, nullptr, 10) : 40960;
	return MaxValue<uint64_t>(mb, 1) << 20;
}

//! Materialise the walks of stream.parts[stream.part] (caller holds the stream's lock or is single-threaded).
void PhysicalGGPathExpand::MaterialisePart(GGExpandStream &stream) const {
	if (stream.result) {
		gg_result_destroy(stream.result);
		stream.result = nullptr;
	}
	auto &part = graph->Part(stream.graph_part);
	lock_guard<mutex> device_guard(part.lock);
	const auto range = stream.parts[stream.part];
	if (stream.by_middle) {
		// (no stats: the counting expansion in front of the rows would only produce a digest nobody reads here)
		GGGraph::Check(gg_expand_khop_mid_result(part.ctx, part.csr, range.first, range.second, k_min, nullptr,
		                                         &stream.result),
		               "gg_expand_khop_mid_result");
		memset(&stream.stats, 0, sizeof(stream.stats));
		for (int h = k_min; h <= 2; h++) {
			GGGraph::Check(gg_result_rows(stream.result, h, &stream.stats.rows[h]), "gg_result_rows");
		}
	} else if (all_sources) {
		GGGraph::Check(gg_expand_khop_range(part.ctx, part.csr, range.first, range.second, k_min, k_max, 1,
		                                    &stream.stats, &stream.result),
		               "gg_expand_khop_range");
	} else {
		GGGraph::Check(gg_expand_khop(part.ctx, part.csr, sources.data() + range.first, range.second - range.first,
		                              k_min, k_max, 1, &stream.stats, &stream.result),
		               "gg_expand_khop");
	}
	stream.hop = k_min;
	stream.offset = 0;
	stream.started = true;
}

unique_ptr<GlobalSourceState> PhysicalGGPathExpand::GetGlobalSourceState(ClientContext &context) const {
	auto state = make_unique<GGExpandGlobalState>();
	lock_guard<mutex> guard(graph->lock);
	if (!graph->csr) {
		throw InternalException("GG_PATH_EXPAND scheduled before the CSR was built");
	}
	const bool product_form = all_sources && k_max == 2; // rows grouped by middle vertex: in(x) x out(x), gg.h
	if (graph->Parts() > 1) {
		// ownership-sharded graph: part p holds the walks whose middle vertex (1-hop: destination) it owns; counts
		// add, digests add, materialised rows concatenate (gg.h: gg_csr_build_shard) — what bench.py's ranks combine
		// with one all-reduce, here across the contexts of one process
		if (!product_form) {
			throw InternalException("GG_PATH_EXPAND over a sharded graph: only the 2-hop walks of all sources (with or without the 1-hop ones)");
		}
		vector<gg_khop_stats> per_part(graph->Parts());
		graph->ForEachPart([&](int p, GGGraph &part) {
			if (!part.csr) {
				throw InternalException("GG_PATH_EXPAND scheduled before the CSR shards were built");
			}
			if (rows_only || !count_only) {
				CountRowsFromDegrees(part.ctx, part.csr, sources, true, k_min, k_max, per_part[p]);
				return;
			}
			GGGraph::Check(gg_expand_khop(part.ctx, part.csr, nullptr, 0, k_min, k_max, 0, &per_part[p], nullptr),
			               "gg_expand_khop");
		});
		memset(&state->stats, 0, sizeof(state->stats));
		for (auto &stats : per_part) {
			for (int h = 0; h <= GG_MAX_HOPS; h++) {
				state->stats.rows[h] += stats.rows[h];
				// (a u32 sum carried in the low half: DESIGN.md "Row digest")
				state->stats.digest[h] = (state->stats.digest[h] + stats.digest[h]) & 0xFFFFFFFFull;
			}
			state->stats.traversed_edges += stats.traversed_edges;
			state->stats.frontier_entries += stats.frontier_entries;
		}
		state->hop = k_min;
		if (count_only) {
			return move(state);
		}
		uint64_t total = 0;
		for (int p = 0; p < graph->Parts(); p++) {
			auto stream = make_unique<GGExpandStream>();
			stream->graph_part = p;
			stream->by_middle = true;
			PlanMiddleParts(*stream, graph->Part(p), per_part[p]);
			state->streams.push_back(move(stream));
			for (int h = k_min; h <= k_max; h++) {
				total += per_part[p].rows[h];
			}
		}
		state->max_threads = MaxValue<idx_t>(1, total / GGResultSlab::SLAB_ROWS);
		return move(state);
	}
	if (rows_only) {
		// count(*) over the join chain: the answer is a sum of degree products, no walk is formed
		CountRowsFromDegrees(graph->ctx, graph->csr, sources, all_sources, k_min, k_max, state->stats);
		state->hop = k_min;
		return move(state);
	}
	// count first: cheap (nothing is written), and it tells how much HBM the walks would take — from degrees when
	// the rows are wanted (their digest is nobody's business then), by the counting expansion for gg_path_count
	if (count_only) {
		GGGraph::Check(gg_expand_khop(graph->ctx, graph->csr, all_sources ? nullptr : SourceIds(sources), sources.size(),
		                              k_min, k_max, 0, &state->stats, nullptr),
		               "gg_expand_khop");
		state->hop = k_min;
		return move(state);
	}
	CountRowsFromDegrees(graph->ctx, graph->csr, sources, all_sources, k_min, k_max, state->stats);
	state->hop = k_min;
	uint64_t total = 0, bytes = 0;
	for (int h = k_min; h <= k_max; h++) {
		total += state->stats.rows[h];
		bytes += state->stats.rows[h] * (uint64_t)(h + 1) * sizeof(int64_t);
	}
	auto stream = make_unique<GGExpandStream>();
	if (product_form) {
		stream->by_middle = true;
		PlanMiddleParts(*stream, *graph, state->stats);
	} else {
		// The reference streams a join result of any size; so must its replacement: beyond the budget the
		// sources are split into parts (twice as many as strictly needed: the split is balanced on 2-hop work,
		// not on output bytes) that are expanded, handed out and freed one after the other.
		uint64_t V = 0;
		GGGraph::Check(gg_csr_info(graph->csr, &V, nullptr, nullptr), "gg_csr_info");
		const uint64_t units = all_sources ? V : sources.size();
		uint64_t n_parts = bytes > ResultBudgetBytes() ? 2 * ((bytes + ResultBudgetBytes() - 1) / ResultBudgetBytes()) : 1;
		n_parts = MaxValue<uint64_t>(1, MinValue<uint64_t>(n_parts, MaxValue<uint64_t>(units, 1)));
		if (n_parts == 1 || units == 0) {
			stream->parts.emplace_back(0, units);
		} else if (all_sources) {
			vector<uint64_t> bounds(n_parts + 1);
			GGGraph::Check(gg_khop_partition(graph->ctx, graph->csr, (int)n_parts, bounds.data()), "gg_khop_partition");
			for (uint64_t i = 0; i < n_parts; i++) {
				if (bounds[i + 1] > bounds[i]) {
					stream->parts.emplace_back(bounds[i], bounds[i + 1]);
				}
			}
		} else {
			for (uint64_t i = 0; i < n_parts; i++) {
				const uint64_t lo = units * i / n_parts, hi = units * (i + 1) / n_parts;
				if (hi > lo) {
					stream->parts.emplace_back(lo, hi);
				}
			}
		}
	}
	state->streams.push_back(move(stream));
	state->max_threads = MaxValue<idx_t>(1, total / GGResultSlab::SLAB_ROWS);
	return move(state);
}

//! Middle-vertex ranges of near-equal product work whose rows fit the budget (gg_khop_partition_mid): the parts of a
//! 2-hop result from every vertex, each produced by the product kernel (k_mat_mid2) like the whole would be.
void PhysicalGGPathExpand::PlanMiddleParts(GGExpandStream &stream, GGGraph &part, const gg_khop_stats &stats) const {
	uint64_t bytes = 0, V = 0;
	for (int h = k_min; h <= k_max; h++) {
		bytes += stats.rows[h] * (uint64_t)(h + 1) * sizeof(int64_t);
	}
	GGGraph::Check(gg_csr_info(part.csr, &V, nullptr, nullptr), "gg_csr_info");
	uint64_t n_parts = (bytes + ResultBudgetBytes() - 1) / ResultBudgetBytes();
	n_parts = MaxValue<uint64_t>(1, MinValue<uint64_t>(n_parts, MaxValue<uint64_t>(V, 1)));
	if (n_parts == 1 || V == 0) {
		stream.parts.emplace_back(0, V);
		return;
	}
	// (the split balances work, not bytes: a quarter more parts keep the largest under the budget)
	n_parts += (n_parts + 3) / 4;
	vector<uint64_t> bounds(n_parts + 1);
	GGGraph::Check(gg_khop_partition_mid(part.ctx, part.csr, (int)n_parts, bounds.data()), "gg_khop_partition_mid");
	for (uint64_t i = 0; i < n_parts; i++) {
		if (bounds[i + 1] > bounds[i]) {
			stream.parts.emplace_back(bounds[i], bounds[i + 1]);
		}
	}
}

unique_ptr<LocalSourceState> PhysicalGGPathExpand::GetLocalSourceState(ExecutionContext &context,
                                                                       GlobalSourceState &gstate) const {
	return make_unique<GGExpandLocalState>(graph);
}

void PhysicalGGPathExpand::GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate_p,
                                   LocalSourceState &lstate) const {
	auto &gstate = (GGExpandGlobalState &)gstate_p;
	if (count_only) {
		lock_guard<mutex> guard(gstate.lock);
		if (gstate.hop > k_max) {
			return; // empty chunk: exhausted (pipeline_executor.cpp:55-58)
		}
		idx_t n = 0;
		for (int h = gstate.hop; h <= k_max; h++, n++) {
			FlatVector::GetData<int32_t>(chunk.data[0])[n] = h;
			FlatVector::GetData<int64_t>(chunk.data[1])[n] = (int64_t)gstate.stats.rows[h];
			FlatVector::GetData<int64_t>(chunk.data[2])[n] = (int64_t)gstate.stats.digest[h];
			FlatVector::GetData<int64_t>(chunk.data[3])[n] = (int64_t)gstate.stats.traversed_edges;
		}
		gstate.hop = k_max + 1;
		chunk.SetCardinality(n);
		return;
	}
	if (context.client.interrupted) { // cancellation is polled between device calls
		throw InterruptException();
	}
	auto &slab = (GGExpandLocalState &)lstate;
	const idx_t n_streams = gstate.streams.size();
	if (slab.stream == INVALID_INDEX) {
		slab.stream = gstate.next_stream++ % n_streams;
	}
	idx_t exhausted = 0; // streams found done in a row
	while (slab.pos >= slab.rows) { // claim the next rows of the next non-empty hop-length table of some stream
		if (exhausted >= n_streams) {
			return; // every part of every stream handed out
		}
		auto &stream = *gstate.streams[slab.stream];
		idx_t offset, want;
		gg_result *result = nullptr;
		{
			lock_guard<mutex> guard(stream.lock);
			while (!stream.done) {
				if (!stream.started) {
					if (stream.parts.empty()) {
						stream.done = true;
						break;
					}
					MaterialisePart(stream);
				}
				while (stream.hop <= k_max && stream.offset >= stream.stats.rows[stream.hop]) {
					stream.hop++;
					stream.offset = 0;
				}
				if (stream.hop <= k_max) {
					break;
				}
				if (stream.part + 1 >= stream.parts.size()) {
					stream.done = true;
					break;
				}
				// this part is claimed completely: wait for the fetches still reading it, then replace it
				while (stream.fetching.load() != 0) {
					std::this_thread::yield();
				}
				stream.part++;
				MaterialisePart(stream);
			}
			if (!stream.done) {
				slab.table = stream.hop;
				offset = stream.offset;
				want = MinValue<idx_t>(GGResultSlab::SLAB_ROWS, stream.stats.rows[stream.hop] - stream.offset);
				stream.offset += want;
				result = stream.result;
				stream.fetching++;
			}
		}
		if (!result) { // this stream has nothing left: try the next one
			slab.stream = (slab.stream + 1) % n_streams;
			exhausted++;
			continue;
		}
		exhausted = 0;
		uint32_t got = 0;
		int rc;
		{
			FetchClaim claim(stream.fetching, true); // released also when slab.Columns() throws (pinned allocation)
			rc = gg_result_fetch(result, slab.table, offset, (uint32_t)want, slab.Columns(slab.table + 1), &got);
		}
		GGGraph::Check(rc, "gg_result_fetch");
		slab.rows = got;
		slab.pos = 0;
	}
	const int hop = slab.table;
	const idx_t n = MinValue<idx_t>(STANDARD_VECTOR_SIZE, slab.rows - slab.pos);
	for (int c = 0; c <= hop; c++) {
		memcpy(FlatVector::GetData<int64_t>(chunk.data[1 + c]), slab.column[c] + slab.pos, n * sizeof(int64_t));
	}
	slab.pos += n;
	auto hops = FlatVector::GetData<int32_t>(chunk.data[0]);
	for (idx_t i = 0; i < n; i++) {
		hops[i] = hop;
	}
	for (int c = hop + 1; c <= k_max; c++) { // shorter walks: trailing vertices are NULL
		chunk.data[1 + c].SetVectorType(VectorType::CONSTANT_VECTOR);
		ConstantVector::SetNull(chunk.data[1 + c], true);
	}
	chunk.SetCardinality(n);
}

//===--------------------------------------------------------------------===//
// Generic single-key inner join: build side sunk into a device index, probe side streamed through it
//===--------------------------------------------------------------------===//
class GGKeyJoinGlobalState : public GlobalSinkState {
public:
	shared_ptr<GGGraph> graph;                 // context + the index keyed on the build key (edge rows key -> key, rowid)
	unique_ptr<PhysicalGGEdgeSink> sink;       // does the staging and the build
	unique_ptr<GlobalSinkState> sink_state;
};

class GGKeyJoinOperatorState : public OperatorState {
public:
	~GGKeyJoinOperatorState() override {
		if (result) {
			gg_result_destroy(result);
		}
	}
	gg_result *result = nullptr; // matches of the probe chunk in hand: (position among its valid keys, build rowid)
	idx_t matches = 0, offset = 0;
	vector<sel_t> valid_rows;     // position among the valid keys -> row of the probe chunk
	vector<vector<int64_t>> scratch;
	vector<const int64_t *> keys;
	vector<int64_t> positions;    // fetched slice
};

PhysicalGGKeyJoin::PhysicalGGKeyJoin(vector<LogicalType> types, unique_ptr<PhysicalOperator> probe,
                                     unique_ptr<PhysicalOperator> build_scan, idx_t probe_key_p,
                                     vector<idx_t> probe_columns_p, TableCatalogEntry *build_table_p,
                                     column_t build_key_p, vector<column_t> build_columns_p, idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::HASH_JOIN, move(types), estimated_cardinality), probe_key(probe_key_p),
      probe_columns(move(probe_columns_p)), build_table(build_table_p), build_key(build_key_p),
      build_columns(move(build_columns_p)) {
	children.push_back(move(probe));
	children.push_back(move(build_scan));
}

string PhysicalGGKeyJoin::ParamsToString() const {
	return "INNER\n" + build_table->name + "." + build_table->columns[build_key].name + " = #" + to_string(probe_key);
}

unique_ptr<GlobalSinkState> PhysicalGGKeyJoin::GetGlobalSinkState(ClientContext &context) const {
	auto state = make_unique<GGKeyJoinGlobalState>();
	state->graph = make_shared<GGGraph>(0, true /* the matches are handed back as rowids */);
	// input chunks: (key, key, rowid) — an edge table whose rows point from their key to their key; the vertex set is
	// the distinct keys (derive_vertices)
	state->sink = make_unique<PhysicalGGEdgeSink>(state->graph, children[1]->types, 0, false, true);
	state->sink_state = state->sink->GetGlobalSinkState(context);
	return move(state);
}

unique_ptr<LocalSinkState> PhysicalGGKeyJoin::GetLocalSinkState(ExecutionContext &context) const {
	auto &gstate = (GGKeyJoinGlobalState &)*sink_state;
	return gstate.sink->GetLocalSinkState(context);
}

SinkResultType PhysicalGGKeyJoin::Sink(ExecutionContext &context, GlobalSinkState &gstate_p, LocalSinkState &lstate,
                                       DataChunk &input) const {
	auto &gstate = (GGKeyJoinGlobalState &)gstate_p;
	return gstate.sink->Sink(context, *gstate.sink_state, lstate, input);
}

void PhysicalGGKeyJoin::Combine(ExecutionContext &context, GlobalSinkState &gstate_p, LocalSinkState &lstate) const {
	auto &gstate = (GGKeyJoinGlobalState &)gstate_p;
	gstate.sink->Combine(context, *gstate.sink_state, lstate);
}

SinkFinalizeType PhysicalGGKeyJoin::Finalize(Pipeline &pipeline, Event &event, ClientContext &context,
                                             GlobalSinkState &gstate_p) const {
	auto &gstate = (GGKeyJoinGlobalState &)gstate_p;
	return gstate.sink->Finalize(pipeline, event, context, *gstate.sink_state);
}

unique_ptr<OperatorState> PhysicalGGKeyJoin::GetOperatorState(ClientContext &context) const {
	return make_unique<GGKeyJoinOperatorState>();
}

OperatorResultType PhysicalGGKeyJoin::Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
                                              OperatorState &state_p) const {
	auto &state = (GGKeyJoinOperatorState &)state_p;
	auto &gstate = (GGKeyJoinGlobalState &)*sink_state;
	auto &graph = *gstate.graph;
	if (!state.result) {
		// a new probe chunk: its valid keys go to the device, the matches stay there until they are handed out
		if (!graph.csr || input.size() == 0) {
			return OperatorResultType::NEED_MORE_INPUT;
		}
		// rows whose key is NULL join nothing (JoinHashTable::PrepareKeys, join_hashtable.cpp:126-148)
		VectorData kdata;
		input.data[probe_key].Orrify(input.size(), kdata);
		const bool narrow = input.data[probe_key].GetType().InternalType() == PhysicalType::INT32;
		state.valid_rows.clear();
		state.scratch.resize(1);
		state.scratch[0].clear();
		for (idx_t r = 0; r < input.size(); r++) {
			const auto idx = kdata.sel->get_index(r);
			if (!kdata.validity.RowIsValid(idx)) {
				continue;
			}
			state.valid_rows.push_back((sel_t)r);
			state.scratch[0].push_back(narrow ? (int64_t)((const int32_t *)kdata.data)[idx] : ((const int64_t *)kdata.data)[idx]);
		}
		if (state.valid_rows.empty()) {
			return OperatorResultType::NEED_MORE_INPUT;
		}
		uint64_t matches = 0;
		{
			lock_guard<mutex> guard(graph.lock); // (device calls on one context are serialised, gg.h)
			GGGraph::Check(gg_join_probe(graph.ctx, graph.csr, state.scratch[0].data(), state.scratch[0].size(), &matches,
			                             &state.result),
			               "gg_join_probe");
		}
		state.matches = matches;
		state.offset = 0;
		if (matches == 0) {
			gg_result_destroy(state.result);
			state.result = nullptr;
			return OperatorResultType::NEED_MORE_INPUT;
		}
	}
	if (context.client.interrupted) {
		throw InterruptException();
	}
	// the next <= 1024 matches: (position, rowid)
	Vector rowids(LOGICAL_ROW_TYPE);
	state.positions.resize(STANDARD_VECTOR_SIZE);
	int64_t *cols[2] = {state.positions.data(), (int64_t *)FlatVector::GetData<row_t>(rowids)};
	uint32_t n = 0;
	{
		lock_guard<mutex> guard(graph.lock);
		GGGraph::Check(gg_result_fetch(state.result, 1, state.offset, STANDARD_VECTOR_SIZE, cols, &n), "gg_result_fetch");
	}
	state.offset += n;
	SelectionVector sel(STANDARD_VECTOR_SIZE);
	auto ids = FlatVector::GetData<row_t>(rowids);
	const idx_t build_rows = build_table->storage->GetTotalRows();
	for (uint32_t r = 0; r < n; r++) {
		if ((uint64_t)state.positions[r] >= state.valid_rows.size()) {
			// (never a wrong row quietly: whatever came back is not a match of this chunk)
			throw InternalException("GG_KEY_JOIN: a probe position outside the chunk came back from the device index");
		}
		sel.set_index(r, state.valid_rows[(idx_t)state.positions[r]]);
		if (ids[r] >= 0 && ids[r] < MAX_ROW_ID && (idx_t)ids[r] >= build_rows) {
			throw InternalException("GG_KEY_JOIN: a rowid past the end of the build table came back from the device index");
		}
		if (ids[r] < 0 || ids[r] >= MAX_ROW_ID) {
			throw NotImplementedException("GG_KEY_JOIN: build rows that this transaction has not committed yet cannot be "
			                              "fetched by rowid (PRAGMA disable_gpu_joins for this statement)");
		}
	}
	// probe columns: the chunk's rows by position (dictionary vectors over the input, as result.Slice(left, sel) in
	// ScanStructure::NextInnerJoin, join_hashtable.cpp:466); build columns: fetched by rowid
	const idx_t left_columns = probe_columns.size();
	for (idx_t c = 0; c < left_columns; c++) {
		chunk.data[c].Slice(input.data[probe_columns[c]], sel, n);
	}
	vector<column_t> fetch_ids;
	vector<LogicalType> fetch_types;
	for (auto column : build_columns) {
		if (column != COLUMN_IDENTIFIER_ROW_ID) {
			fetch_ids.push_back(column);
			fetch_types.push_back(build_table->columns[column].type);
		}
	}
	DataChunk fetched;
	if (!fetch_ids.empty()) {
		fetched.Initialize(fetch_types);
		ColumnFetchState fetch_state;
		auto &transaction = Transaction::GetTransaction(context.client);
		build_table->storage->Fetch(transaction, fetched, fetch_ids, rowids, n, fetch_state);
		if (fetched.size() != n) {
			throw InternalException("GG_KEY_JOIN: a build row of the join is not visible to the statement any more");
		}
	}
	idx_t f = 0;
	for (idx_t c = 0; c < build_columns.size(); c++) {
		if (build_columns[c] == COLUMN_IDENTIFIER_ROW_ID) {
			chunk.data[left_columns + c].Reference(rowids);
		} else {
			chunk.data[left_columns + c].Reference(fetched.data[f++]);
		}
	}
	chunk.SetCardinality(n);
	if (state.offset >= state.matches) {
		gg_result_destroy(state.result);
		state.result = nullptr;
		return OperatorResultType::NEED_MORE_INPUT;
	}
	return OperatorResultType::HAVE_MORE_OUTPUT;
}

//===--------------------------------------------------------------------===//
// Filtered paths source (ConnectedSegments)
//===--------------------------------------------------------------------===//
class GGFilteredGlobalState : public GlobalSourceState {
public:
	~GGFilteredGlobalState() override {
		if (result) {
			gg_result_destroy(result);
		}
	}
	gg_result *result = nullptr;
	idx_t rows = 0;
	idx_t offset = 0;
};

static vector<LogicalType> BigintColumns(idx_t n) {
	return vector<LogicalType>(n, LogicalType::BIGINT);
}

PhysicalGGFilteredPaths::PhysicalGGFilteredPaths(shared_ptr<GGGraph> graph_p, int hops_p, vector<int64_t> sources_p,
                                                 idx_t estimated_cardinality, bool all_sources_p)
    : PhysicalOperator(PhysicalOperatorType::INVALID, BigintColumns(hops_p + 2), estimated_cardinality),
      graph(move(graph_p)), hops(hops_p), sources(move(sources_p)), all_sources(all_sources_p) {
}

unique_ptr<GlobalSourceState> PhysicalGGFilteredPaths::GetGlobalSourceState(ClientContext &context) const {
	auto state = make_unique<GGFilteredGlobalState>();
	lock_guard<mutex> guard(graph->lock);
	if (!graph->csr || !graph->filter_csr) {
		throw InternalException("GG_FILTERED_PATHS scheduled before both CSRs were built");
	}
	gg_khop_stats stats;
	gg_result *paths = nullptr;
	GGGraph::Check(gg_expand_khop_result(graph->ctx, graph->csr, all_sources ? nullptr : SourceIds(sources),
	                                     all_sources ? 0 : sources.size(), hops, hops, &stats, &paths),
	               "gg_expand_khop_result");
	int rc = gg_result_filter_common_neighbour(graph->ctx, paths, hops, graph->filter_csr, &state->result);
	gg_result_destroy(paths);
	GGGraph::Check(rc, "gg_result_filter_common_neighbour");
	uint64_t n = 0;
	GGGraph::Check(gg_result_rows(state->result, hops + 1, &n), "gg_result_rows");
	state->rows = n;
	return move(state);
}

void PhysicalGGFilteredPaths::GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate_p,
                                      LocalSourceState &lstate) const {
	auto &gstate = (GGFilteredGlobalState &)gstate_p;
	if (gstate.offset >= gstate.rows) {
		return;
	}
	int64_t *cols[GG_MAX_HOPS + 2];
	for (int c = 0; c <= hops + 1; c++) {
		cols[c] = FlatVector::GetData<int64_t>(chunk.data[c]);
	}
	uint32_t n = 0;
	{
		lock_guard<mutex> guard(graph->lock);
		GGGraph::Check(gg_result_fetch(gstate.result, hops + 1, gstate.offset, STANDARD_VECTOR_SIZE, cols, &n),
		               "gg_result_fetch");
	}
	gstate.offset += n;
	chunk.SetCardinality(n);
}

//===--------------------------------------------------------------------===//
// Walks with payload columns of their edges
//===--------------------------------------------------------------------===//
vector<LogicalType> PhysicalGGPathEdges::OutputTypes(int hops, TableCatalogEntry &edge_table,
                                                     const vector<std::pair<idx_t, column_t>> &payload) {
	vector<LogicalType> types {LogicalType::INTEGER};
	for (int c = 0; c <= hops; c++) {
		types.push_back(LogicalType::BIGINT);
	}
	for (auto &entry : payload) {
		types.push_back(edge_table.columns[entry.second].type);
	}
	return types;
}

PhysicalGGPathEdges::PhysicalGGPathEdges(shared_ptr<GGGraph> graph_p, int hops_p, vector<int64_t> sources_p,
                                         bool all_sources_p, TableCatalogEntry *edge_table_p,
                                         vector<std::pair<idx_t, column_t>> payload_p, idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::INVALID, OutputTypes(hops_p, *edge_table_p, payload_p),
                       estimated_cardinality),
      graph(move(graph_p)), hops(hops_p), sources(move(sources_p)), all_sources(all_sources_p),
      edge_table(edge_table_p), payload(move(payload_p)) {
}

//! State of GG_PATH_EDGES: like the walks without edges, a result beyond the device-memory budget is produced part by
//! part — slices of the source list (all sources: of the vertex ids) expanded, handed out and freed in turn.
class GGPathEdgesGlobalState : public GlobalSourceState {
public:
	~GGPathEdgesGlobalState() override {
		if (result) {
			gg_result_destroy(result);
		}
	}
	gg_result *result = nullptr;
	idx_t rows = 0;
	idx_t offset = 0;
	vector<int64_t> ids; // the sources the parts slice (all sources: every vertex id)
	vector<std::pair<uint64_t, uint64_t>> parts;
	idx_t part = 0;
};

void PhysicalGGPathEdges::MaterialisePart(GlobalSourceState &gstate_p) const {
	auto &state = (GGPathEdgesGlobalState &)gstate_p;
	if (state.result) {
		gg_result_destroy(state.result);
		state.result = nullptr;
	}
	gg_khop_stats stats;
	const auto range = state.parts[state.part];
	const bool whole = all_sources && state.parts.size() == 1;
	GGGraph::Check(gg_expand_khop_edges(graph->ctx, graph->csr, whole ? nullptr : state.ids.data() + range.first,
	                                    whole ? 0 : range.second - range.first, hops, &stats, &state.result),
	               "gg_expand_khop_edges");
	uint64_t n = 0;
	GGGraph::Check(gg_result_rows(state.result, hops, &n), "gg_result_rows");
	state.rows = n;
	state.offset = 0;
}

unique_ptr<GlobalSourceState> PhysicalGGPathEdges::GetGlobalSourceState(ClientContext &context) const {
	auto state = make_unique<GGPathEdgesGlobalState>();
	lock_guard<mutex> guard(graph->lock);
	if (!graph->csr) {
		throw InternalException("GG_PATH_EDGES scheduled before the CSR was built");
	}
	// count first (from degrees: nothing is written) — the walks with their edges take (2 * hops + 1) int64 columns
	// at the last level plus the level tables below it (dense u32 columns: about half as much again)
	uint64_t rows[GG_MAX_HOPS + 1];
	GGGraph::Check(gg_khop_count(graph->ctx, graph->csr, all_sources ? nullptr : SourceIds(sources), sources.size(), 1, hops,
	                             rows),
	               "gg_khop_count");
	uint64_t bytes = rows[hops] * (uint64_t)(2 * hops + 1) * sizeof(int64_t);
	for (int h = 1; h < hops; h++) {
		bytes += rows[h] * (uint64_t)(2 * h + 1) * sizeof(uint32_t);
	}
	const uint64_t budget = ResultBudgetBytes();
	uint64_t V = 0;
	GGGraph::Check(gg_csr_info(graph->csr, &V, nullptr, nullptr), "gg_csr_info");
	const uint64_t units = all_sources ? V : sources.size();
	// (slices of equal size, not of equal work: four times the parts the bytes ask for)
	uint64_t n_parts = bytes > budget ? 4 * ((bytes + budget - 1) / budget) : 1;
	n_parts = MaxValue<uint64_t>(1, MinValue<uint64_t>(n_parts, MaxValue<uint64_t>(units, 1)));
	if (n_parts == 1) {
		state->parts.emplace_back(0, units);
		if (!all_sources) {
			state->ids = sources;
		}
	} else {
		if (all_sources) {
			state->ids.resize(V);
			GGGraph::Check(gg_csr_export(graph->csr, nullptr, nullptr, nullptr, state->ids.data()), "gg_csr_export");
		} else {
			state->ids = sources;
		}
		for (uint64_t i = 0; i < n_parts; i++) {
			const uint64_t lo = units * i / n_parts, hi = units * (i + 1) / n_parts;
			if (hi > lo) {
				state->parts.emplace_back(lo, hi);
			}
		}
	}
	MaterialisePart(*state);
	return move(state);
}

void PhysicalGGPathEdges::GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate_p,
                                  LocalSourceState &lstate) const {
	auto &gstate = (GGPathEdgesGlobalState &)gstate_p;
	while (gstate.offset >= gstate.rows) { // this part is handed out: the next one, if any
		if (gstate.part + 1 >= gstate.parts.size()) {
			return;
		}
		if (context.client.interrupted) {
			throw InterruptException();
		}
		gstate.part++;
		lock_guard<mutex> guard(graph->lock);
		MaterialisePart(gstate);
	}
	int64_t *cols[GG_MAX_HOPS + 1];
	for (int c = 0; c <= hops; c++) {
		cols[c] = FlatVector::GetData<int64_t>(chunk.data[1 + c]);
	}
	vector<Vector> rowids;
	int64_t *ecols[GG_MAX_HOPS + 1];
	for (int j = 0; j < hops; j++) {
		rowids.emplace_back(LOGICAL_ROW_TYPE);
		ecols[j] = (int64_t *)FlatVector::GetData<row_t>(rowids.back());
	}
	uint32_t n = 0, ne = 0;
	{
		lock_guard<mutex> guard(graph->lock);
		GGGraph::Check(gg_result_fetch(gstate.result, hops, gstate.offset, STANDARD_VECTOR_SIZE, cols, &n), "gg_result_fetch");
		GGGraph::Check(gg_result_fetch_edges(gstate.result, hops, gstate.offset, STANDARD_VECTOR_SIZE, ecols, &ne),
		               "gg_result_fetch_edges");
	}
	if (n != ne) {
		throw InternalException("GG_PATH_EDGES: vertex and edge columns out of step");
	}
	gstate.offset += n;
	chunk.data[0].Reference(Value::INTEGER(hops));
	// the payload columns of edge j, all in one fetch by rowid
	auto &transaction = Transaction::GetTransaction(context.client);
	for (int j = 1; j <= hops; j++) {
		vector<column_t> column_ids;
		vector<LogicalType> types;
		vector<idx_t> targets;
		for (idx_t p = 0; p < payload.size(); p++) {
			if ((int)payload[p].first == j) {
				column_ids.push_back(payload[p].second);
				types.push_back(edge_table->columns[payload[p].second].type);
				targets.push_back(2 + hops + p);
			}
		}
		if (column_ids.empty()) {
			continue;
		}
		auto ids = FlatVector::GetData<row_t>(rowids[j - 1]);
		for (uint32_t r = 0; r < n; r++) {
			if (ids[r] < 0 || ids[r] >= MAX_ROW_ID) {
				throw NotImplementedException("GG_PATH_EDGES: edge rows that this transaction has not committed yet cannot be "
				                              "fetched by rowid (PRAGMA disable_gpu_graph for this statement)");
			}
		}
		DataChunk fetched;
		fetched.Initialize(types);
		ColumnFetchState fetch_state;
		edge_table->storage->Fetch(transaction, fetched, column_ids, rowids[j - 1], n, fetch_state);
		if (fetched.size() != n) {
			throw InternalException("GG_PATH_EDGES: an edge row of the walk is not visible to the statement any more");
		}
		for (idx_t c = 0; c < targets.size(); c++) {
			chunk.data[targets[c]].Reference(fetched.data[c]);
		}
	}
	chunk.SetCardinality(n);
}

//===--------------------------------------------------------------------===//
// Distinct walk endpoints source
//===--------------------------------------------------------------------===//
vector<LogicalType> PhysicalGGWalkEndpoints::OutputTypes(int k_max) {
	return BigintColumns(1 + k_max);
}

PhysicalGGWalkEndpoints::PhysicalGGWalkEndpoints(shared_ptr<GGGraph> graph_p, vector<int64_t> sources_p, int k_max_p,
                                                 idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::INVALID, OutputTypes(k_max_p), estimated_cardinality),
      graph(move(graph_p)), sources(move(sources_p)), k_max(k_max_p) {
}

unique_ptr<GlobalSourceState> PhysicalGGWalkEndpoints::GetGlobalSourceState(ClientContext &context) const {
	auto state = make_unique<GGFilteredGlobalState>();
	lock_guard<mutex> guard(graph->lock);
	if (!graph->csr) {
		throw InternalException("GG_WALK_ENDPOINTS scheduled before the CSR was built");
	}
	GGGraph::Check(gg_walk_endpoints(graph->ctx, graph->csr, sources.data(), sources.size(), k_max, &state->result),
	               "gg_walk_endpoints");
	uint64_t n = 0;
	GGGraph::Check(gg_result_rows(state->result, 1, &n), "gg_result_rows");
	state->rows = n;
	return move(state);
}

void PhysicalGGWalkEndpoints::GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate_p,
                                      LocalSourceState &lstate) const {
	auto &gstate = (GGFilteredGlobalState &)gstate_p;
	if (gstate.offset >= gstate.rows) {
		return;
	}
	int64_t masks[STANDARD_VECTOR_SIZE];
	int64_t *cols[2] = {FlatVector::GetData<int64_t>(chunk.data[0]), masks};
	uint32_t n = 0;
	{
		lock_guard<mutex> guard(graph->lock);
		GGGraph::Check(gg_result_fetch(gstate.result, 1, gstate.offset, STANDARD_VECTOR_SIZE, cols, &n),
		               "gg_result_fetch");
	}
	for (int h = 1; h <= k_max; h++) {
		auto flags = FlatVector::GetData<int64_t>(chunk.data[h]);
		for (uint32_t i = 0; i < n; i++) {
			flags[i] = (masks[i] >> h) & 1;
		}
	}
	gstate.offset += n;
	chunk.SetCardinality(n);
}

//===--------------------------------------------------------------------===//
// Shortest path source
//===--------------------------------------------------------------------===//
class GGShortestGlobalState : public GlobalSourceState {
public:
	~GGShortestGlobalState() override {
		if (result) {
			gg_result_destroy(result);
		}
	}
	idx_t MaxThreads() override {
		return max_threads;
	}
	vector<int64_t> uniq;            // the sources, deduplicated, in 64-lane batches
	shared_ptr<const vector<int64_t>> vid; // vertex ids by dense index (the rows come back packed, see RunBatch)
	idx_t batch_base = 0;            // first source of the batch whose rows are in `result`
	gg_result *result = nullptr;     // packed (lane, distance, dense vertex) rows of the current batch, in HBM
	idx_t rows = 0;                  // ... and how many
	std::atomic<idx_t> fetching {0}; // slab fetches still reading `result`
	//! seed rows of sources that are not vertices of the graph (lone_sources): served once, at the end
	vector<int64_t> lone;
	mutex lock;
	idx_t offset = 0; // next unclaimed row of the current batch
	idx_t lone_offset = 0;
	idx_t max_threads = 1;
};

PhysicalGGShortestPath::PhysicalGGShortestPath(shared_ptr<GGGraph> graph_p, vector<int64_t> sources_p,
                                               int max_hops_p, idx_t estimated_cardinality, bool lone_sources_p)
    : PhysicalOperator(PhysicalOperatorType::INVALID,
                       {LogicalType::BIGINT, LogicalType::BIGINT, LogicalType::INTEGER}, estimated_cardinality),
      graph(move(graph_p)), sources(move(sources_p)), max_hops(max_hops_p), lone_sources(lone_sources_p) {
}

//! Run the 64-lane BFS of the batch starting at batch_base; its reached (source, vertex, distance) rows are
//! compacted on the device and stay there until the pipeline threads have fetched them.  One batch is
//! resident at a time: "every person" as seeds at SF100 is 7 000 batches of up to 0.7 GB of rows each.
void PhysicalGGShortestPath::RunBatch(GlobalSourceState &gstate_p) const {
	auto &state = (GGShortestGlobalState &)gstate_p;
	if (state.result) {
		gg_result_destroy(state.result);
		state.result = nullptr;
	}
	const int n = (int)MinValue<idx_t>(GG_BFS_LANES, state.uniq.size() - state.batch_base);
	// one 8-byte word per row (lane << 58 | distance << 32 | dense vertex): a third of the bytes of three
	// id columns over PCIe; the pipeline threads turn lane and dense index back into ids as they unpack
	GGGraph::Check(gg_bfs64_pairs_packed(graph->ctx, graph->csr, state.uniq.data() + state.batch_base, n, max_hops,
	                                     nullptr, &state.result),
	               "gg_bfs64_pairs_packed");
	uint64_t rows = 0;
	GGGraph::Check(gg_result_rows(state.result, 0, &rows), "gg_result_rows");
	state.rows = rows;
	state.offset = 0;
}

unique_ptr<GlobalSourceState> PhysicalGGShortestPath::GetGlobalSourceState(ClientContext &context) const {
	auto state = make_unique<GGShortestGlobalState>();
	lock_guard<mutex> guard(graph->lock);
	if (!graph->csr) {
		throw InternalException("GG_SHORTEST_PATH scheduled before the CSR was built");
	}
	// UNION semantics: a source listed twice yields its rows once
	{
		unordered_set<int64_t> seen;
		for (auto s : sources) {
			if (seen.insert(s).second) {
				state->uniq.push_back(s);
			}
		}
	}
	if (lone_sources && !state->uniq.empty()) {
		// a source that is not a vertex of the graph reaches nothing but keeps its own seed row
		vector<uint32_t> dense(state->uniq.size());
		GGGraph::Check(gg_csr_lookup(graph->ctx, graph->csr, state->uniq.data(), state->uniq.size(), dense.data()),
		               "gg_csr_lookup");
		for (idx_t i = 0; i < state->uniq.size(); i++) {
			if (dense[i] == 0xFFFFFFFFu) {
				state->lone.push_back(state->uniq[i]);
			}
		}
	}
	if (!state->uniq.empty()) {
		state->vid = graph->VertexIds();
		RunBatch(*state);
	}
	// the first batch's size is the only estimate there is of how much the threads will have to drain
	const idx_t batches = (state->uniq.size() + GG_BFS_LANES - 1) / GG_BFS_LANES;
	state->max_threads = MaxValue<idx_t>(1, state->rows * batches / GGResultSlab::SLAB_ROWS);
	return move(state);
}

unique_ptr<LocalSourceState> PhysicalGGShortestPath::GetLocalSourceState(ExecutionContext &context,
                                                                         GlobalSourceState &gstate) const {
	return make_unique<GGResultSlab>(graph);
}

void PhysicalGGShortestPath::GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate_p,
                                     LocalSourceState &lstate) const {
	auto &gstate = (GGShortestGlobalState &)gstate_p;
	auto &slab = (GGResultSlab &)lstate;
	if (context.client.interrupted) {
		throw InterruptException();
	}
	if (slab.pos >= slab.rows) {
		idx_t offset, want;
		gg_result *result;
		{
			lock_guard<mutex> guard(gstate.lock);
			while (gstate.offset >= gstate.rows) { // current batch claimed completely: run the next one
				if (gstate.batch_base + GG_BFS_LANES >= gstate.uniq.size()) {
					if (gstate.uniq.empty() || gstate.lone_offset >= gstate.lone.size()) {
						return;
					}
					// last: the seed rows of the sources that are not vertices
					const idx_t n = MinValue<idx_t>(STANDARD_VECTOR_SIZE, gstate.lone.size() - gstate.lone_offset);
					for (idx_t i = 0; i < n; i++) {
						FlatVector::GetData<int64_t>(chunk.data[0])[i] = gstate.lone[gstate.lone_offset + i];
						FlatVector::GetData<int64_t>(chunk.data[1])[i] = gstate.lone[gstate.lone_offset + i];
						FlatVector::GetData<int32_t>(chunk.data[2])[i] = 0;
					}
					gstate.lone_offset += n;
					chunk.SetCardinality(n);
					return;
				}
				while (gstate.fetching.load() != 0) { // fetches still reading the batch that is about to go
					std::this_thread::yield();
				}
				gstate.batch_base += GG_BFS_LANES;
				lock_guard<mutex> device_guard(graph->lock);
				RunBatch(gstate);
			}
			offset = gstate.offset;
			want = MinValue<idx_t>(GGResultSlab::SLAB_ROWS, gstate.rows - offset);
			gstate.offset += want;
			result = gstate.result;
			slab.table = (int)gstate.batch_base; // lane i of these rows is source uniq[batch_base + i]
			gstate.fetching++;
		}
		uint32_t got = 0;
		int rc;
		{
			FetchClaim claim(gstate.fetching, true);
			rc = gg_result_fetch(result, 0, offset, (uint32_t)want, slab.Columns(1), &got);
		}
		GGGraph::Check(rc, "gg_result_fetch");
		slab.rows = got;
		slab.pos = 0;
		if (got == 0) {
			return;
		}
	}
	const idx_t n = MinValue<idx_t>(STANDARD_VECTOR_SIZE, slab.rows - slab.pos);
	auto start = FlatVector::GetData<int64_t>(chunk.data[0]);
	auto frnd = FlatVector::GetData<int64_t>(chunk.data[1]);
	auto hops = FlatVector::GetData<int32_t>(chunk.data[2]);
	const int64_t *lane_source = gstate.uniq.data() + slab.table;
	const int64_t *vertex_id = gstate.vid->data();
	const int64_t *packed = slab.column[0] + slab.pos;
	for (idx_t i = 0; i < n; i++) { // lane << 58 | distance << 32 | dense vertex index
		const uint64_t w = (uint64_t)packed[i];
		start[i] = lane_source[w >> 58];
		frnd[i] = vertex_id[(uint32_t)w];
		hops[i] = (int32_t)((w >> 32) & 0x3FFFFFFu);
	}
	slab.pos += n;
	chunk.SetCardinality(n);
}

} // namespace duckdb
