// gg_operators.cpp — see gg_operators.hpp.
#include "gg_operators.hpp"

#include "duckdb/common/exception.hpp"
#include "duckdb/common/types/data_chunk.hpp"
#include "duckdb/common/types/vector.hpp"
#include "duckdb/main/client_context.hpp"

namespace duckdb {

//===--------------------------------------------------------------------===//
// GGGraph
//===--------------------------------------------------------------------===//
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

GGGraph::GGGraph(int device_p) : device(device_p) {
	ctx = g_context_pool.Acquire(device);
	if (ctx) {
		Check(gg_staging_clear(ctx), "gg_staging_clear");
	} else {
		Check(gg_ctx_create(device, &ctx), "gg_ctx_create");
	}
	// none of the operators below returns edge ids: build without the edge-rowid payload, as the
	// reference's hash-join build side carries only the columns the query references
	Check(gg_ctx_set_edge_rowid(ctx, 0), "gg_ctx_set_edge_rowid");
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
// Sinks
//===--------------------------------------------------------------------===//
class GGSinkGlobalState : public GlobalSinkState {
public:
	std::atomic<idx_t> rows {0};
};

class GGSinkLocalState : public LocalSinkState {
public:
	vector<vector<int64_t>> columns; // per-thread conversion buffers, reused across chunks
	vector<const int64_t *> keys;    // what is handed to gg_*_append: the chunk's own vectors or `columns`
};

PhysicalGGVertexSink::PhysicalGGVertexSink(shared_ptr<GGGraph> graph_p, vector<LogicalType> types,
                                           idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::INVALID, move(types), estimated_cardinality), graph(move(graph_p)) {
}

unique_ptr<GlobalSinkState> PhysicalGGVertexSink::GetGlobalSinkState(ClientContext &context) const {
	GGGraph::Check(gg_staging_clear(graph->ctx), "gg_staging_clear");
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
	GGGraph::Check(gg_vertices_append(graph->ctx, lstate.keys[0], n), "gg_vertices_append");
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
                                       idx_t estimated_cardinality, bool as_filter_p, bool derive_vertices_p)
    : PhysicalOperator(PhysicalOperatorType::INVALID, move(types), estimated_cardinality), graph(move(graph_p)),
      as_filter(as_filter_p), derive_vertices(derive_vertices_p) {
}

unique_ptr<GlobalSinkState> PhysicalGGEdgeSink::GetGlobalSinkState(ClientContext &context) const {
	if (as_filter) {
		// second edge table over the same staged vertices: drop the first table's staged rows only
		GGGraph::Check(gg_staging_clear_edges(graph->ctx), "gg_staging_clear_edges");
	} else if (derive_vertices) {
		// no vertex sink ran before this one: start from empty staging
		GGGraph::Check(gg_staging_clear(graph->ctx), "gg_staging_clear");
	}
	return make_unique<GGSinkGlobalState>();
}

unique_ptr<LocalSinkState> PhysicalGGEdgeSink::GetLocalSinkState(ExecutionContext &context) const {
	return make_unique<GGSinkLocalState>();
}

SinkResultType PhysicalGGEdgeSink::Sink(ExecutionContext &context, GlobalSinkState &gstate_p, LocalSinkState &lstate_p,
                                        DataChunk &input) const {
	auto &gstate = (GGSinkGlobalState &)gstate_p;
	auto &lstate = (GGSinkLocalState &)lstate_p;
	const bool has_rowid = input.ColumnCount() >= 3;
	idx_t n = has_rowid ? GGKeyColumns(input, {0, 1, 2}, lstate.columns, lstate.keys)
	                    : GGKeyColumns(input, {0, 1}, lstate.columns, lstate.keys);
	GGGraph::Check(gg_edges_append(graph->ctx, lstate.keys[0], lstate.keys[1], has_rowid ? lstate.keys[2] : nullptr, n),
	               "gg_edges_append");
	gstate.rows += n;
	return SinkResultType::NEED_MORE_INPUT;
}

void PhysicalGGEdgeSink::Combine(ExecutionContext &context, GlobalSinkState &gstate, LocalSinkState &lstate) const {
}

SinkFinalizeType PhysicalGGEdgeSink::Finalize(Pipeline &pipeline, Event &event, ClientContext &context,
                                              GlobalSinkState &gstate) const {
	// single-threaded, after every Sink/Combine (physical_operator.hpp:145-147): build the index
	lock_guard<mutex> guard(graph->lock);
	gg_csr *&target = as_filter ? graph->filter_csr : graph->csr;
	if (target) {
		gg_csr_destroy(target);
		target = nullptr;
	}
	if (derive_vertices) {
		GGGraph::Check(gg_vertices_from_edges(graph->ctx, nullptr), "gg_vertices_from_edges");
	}
	GGGraph::Check(gg_csr_build(graph->ctx, &target), "gg_csr_build");
	return SinkFinalizeType::READY;
}

//===--------------------------------------------------------------------===//
// Path expansion source
//===--------------------------------------------------------------------===//
class GGExpandGlobalState : public GlobalSourceState {
public:
	~GGExpandGlobalState() override {
		if (result) {
			gg_result_destroy(result);
		}
	}
	idx_t MaxThreads() override {
		return max_threads;
	}

	gg_khop_stats stats;
	gg_result *result = nullptr;
	// scan position: (current hop length, row offset inside it); GetData serves <=1024-row chunks out of a
	// host slab that is refilled from the device result SLAB_ROWS rows at a time (one copy per column
	// instead of one per chunk)
	static constexpr idx_t SLAB_ROWS = 1u << 18;
	mutex lock;
	int hop = 0;
	idx_t offset = 0;      // next device row of `hop` to fetch
	vector<int64_t> slab[GG_MAX_HOPS + 1];
	int slab_hop = 0;
	idx_t slab_rows = 0, slab_pos = 0;
	idx_t max_threads = 1;
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
                                           idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::INVALID, OutputTypes(k_max_p, count_only_p), estimated_cardinality),
      graph(move(graph_p)), k_min(k_min_p), k_max(k_max_p), count_only(count_only_p), sources(move(sources_p)),
      all_sources(all_sources_p) {
}

unique_ptr<GlobalSourceState> PhysicalGGPathExpand::GetGlobalSourceState(ClientContext &context) const {
	auto state = make_unique<GGExpandGlobalState>();
	lock_guard<mutex> guard(graph->lock);
	if (!graph->csr) {
		throw InternalException("GG_PATH_EXPAND scheduled before the CSR was built");
	}
	GGGraph::Check(gg_expand_khop(graph->ctx, graph->csr, all_sources ? nullptr : sources.data(), sources.size(), k_min,
	                              k_max, count_only ? 0 : 1, &state->stats, count_only ? nullptr : &state->result),
	               "gg_expand_khop");
	state->hop = k_min;
	idx_t total = 0;
	for (int h = k_min; h <= k_max; h++) {
		total += state->stats.rows[h];
	}
	state->max_threads = count_only ? 1 : MaxValue<idx_t>(1, total / (STANDARD_VECTOR_SIZE * 64));
	return move(state);
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
	lock_guard<mutex> guard(gstate.lock);
	if (gstate.slab_pos >= gstate.slab_rows) { // refill the slab from the next non-empty hop-length table
		while (gstate.hop <= k_max && gstate.offset >= gstate.stats.rows[gstate.hop]) {
			gstate.hop++;
			gstate.offset = 0;
		}
		if (gstate.hop > k_max) {
			return;
		}
		const idx_t want = MinValue<idx_t>(GGExpandGlobalState::SLAB_ROWS, gstate.stats.rows[gstate.hop] - gstate.offset);
		int64_t *cols[GG_MAX_HOPS + 1];
		for (int c = 0; c <= gstate.hop; c++) {
			gstate.slab[c].resize(want);
			cols[c] = gstate.slab[c].data();
		}
		uint32_t got = 0;
		{
			lock_guard<mutex> device_guard(graph->lock);
			GGGraph::Check(gg_result_fetch(gstate.result, gstate.hop, gstate.offset, (uint32_t)want, cols, &got),
			               "gg_result_fetch");
		}
		gstate.slab_hop = gstate.hop;
		gstate.slab_rows = got;
		gstate.slab_pos = 0;
		gstate.offset += got;
		if (got == 0) {
			return;
		}
	}
	const int hop = gstate.slab_hop;
	const idx_t n = MinValue<idx_t>(STANDARD_VECTOR_SIZE, gstate.slab_rows - gstate.slab_pos);
	for (int c = 0; c <= hop; c++) {
		memcpy(FlatVector::GetData<int64_t>(chunk.data[1 + c]), gstate.slab[c].data() + gstate.slab_pos,
		       n * sizeof(int64_t));
	}
	gstate.slab_pos += n;
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
                                                 idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::INVALID, BigintColumns(hops_p + 2), estimated_cardinality),
      graph(move(graph_p)), hops(hops_p), sources(move(sources_p)) {
}

unique_ptr<GlobalSourceState> PhysicalGGFilteredPaths::GetGlobalSourceState(ClientContext &context) const {
	auto state = make_unique<GGFilteredGlobalState>();
	lock_guard<mutex> guard(graph->lock);
	if (!graph->csr || !graph->filter_csr) {
		throw InternalException("GG_FILTERED_PATHS scheduled before both CSRs were built");
	}
	gg_khop_stats stats;
	gg_result *paths = nullptr;
	GGGraph::Check(gg_expand_khop_result(graph->ctx, graph->csr, sources.data(), sources.size(), hops, hops, &stats,
	                                     &paths),
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
// Shortest path source
//===--------------------------------------------------------------------===//
class GGShortestGlobalState : public GlobalSourceState {
public:
	vector<int64_t> start, frnd, hop; // (source, vertex, distance) rows of all batches
	idx_t offset = 0;
};

PhysicalGGShortestPath::PhysicalGGShortestPath(shared_ptr<GGGraph> graph_p, vector<int64_t> sources_p,
                                               int max_hops_p, idx_t estimated_cardinality, bool lone_sources_p)
    : PhysicalOperator(PhysicalOperatorType::INVALID,
                       {LogicalType::BIGINT, LogicalType::BIGINT, LogicalType::INTEGER}, estimated_cardinality),
      graph(move(graph_p)), sources(move(sources_p)), max_hops(max_hops_p), lone_sources(lone_sources_p) {
}

unique_ptr<GlobalSourceState> PhysicalGGShortestPath::GetGlobalSourceState(ClientContext &context) const {
	auto state = make_unique<GGShortestGlobalState>();
	lock_guard<mutex> guard(graph->lock);
	if (!graph->csr) {
		throw InternalException("GG_SHORTEST_PATH scheduled before the CSR was built");
	}
	// UNION semantics: a source listed twice yields its rows once
	vector<int64_t> uniq;
	{
		unordered_set<int64_t> seen;
		for (auto s : sources) {
			if (seen.insert(s).second) {
				uniq.push_back(s);
			}
		}
	}
	for (idx_t base = 0; base < uniq.size(); base += GG_BFS_LANES) { // 64 bit lanes per batch
		if (context.interrupted) {
			throw InterruptException();
		}
		const int n = (int)MinValue<idx_t>(GG_BFS_LANES, uniq.size() - base);
		// the reached (source, vertex, distance) rows are compacted on the device and come back in one copy
		gg_result *pairs = nullptr;
		GGGraph::Check(gg_bfs64_pairs(graph->ctx, graph->csr, uniq.data() + base, n, max_hops, nullptr, &pairs),
		               "gg_bfs64_pairs");
		uint64_t rows = 0;
		int rc = gg_result_rows(pairs, 2, &rows);
		const idx_t before = state->start.size();
		if (rc == GG_OK && rows) {
			state->start.resize(before + rows);
			state->frnd.resize(before + rows);
			state->hop.resize(before + rows);
			int64_t *cols[3] = {state->start.data() + before, state->frnd.data() + before, state->hop.data() + before};
			for (uint64_t done = 0; rc == GG_OK && done < rows;) {
				int64_t *at[3] = {cols[0] + done, cols[1] + done, cols[2] + done};
				uint32_t got = 0;
				rc = gg_result_fetch(pairs, 2, done, (uint32_t)MinValue<uint64_t>(rows - done, 1u << 30), at, &got);
				done += got;
			}
		}
		gg_result_destroy(pairs);
		GGGraph::Check(rc, "gg_result_fetch");
		if (lone_sources) {
			// a source that is not a vertex of the graph reached nothing: only its own seed row.  Every
			// source that IS a vertex has its (s, s, 0) row among the fetched ones.
			unordered_set<int64_t> known;
			for (idx_t r = before; r < state->start.size(); r++) {
				if (state->hop[r] == 0) {
					known.insert(state->start[r]);
				}
			}
			for (int i = 0; i < n; i++) {
				if (!known.count(uniq[base + i])) {
					state->start.push_back(uniq[base + i]);
					state->frnd.push_back(uniq[base + i]);
					state->hop.push_back(0);
				}
			}
		}
	}
	return move(state);
}

void PhysicalGGShortestPath::GetData(ExecutionContext &context, DataChunk &chunk, GlobalSourceState &gstate_p,
                                     LocalSourceState &lstate) const {
	auto &gstate = (GGShortestGlobalState &)gstate_p;
	idx_t n = MinValue<idx_t>(STANDARD_VECTOR_SIZE, gstate.start.size() - gstate.offset);
	if (n == 0) {
		return;
	}
	memcpy(FlatVector::GetData<int64_t>(chunk.data[0]), gstate.start.data() + gstate.offset, n * sizeof(int64_t));
	memcpy(FlatVector::GetData<int64_t>(chunk.data[1]), gstate.frnd.data() + gstate.offset, n * sizeof(int64_t));
	auto hops = FlatVector::GetData<int32_t>(chunk.data[2]);
	for (idx_t i = 0; i < n; i++) {
		hops[i] = (int32_t)gstate.hop[gstate.offset + i];
	}
	gstate.offset += n;
	chunk.SetCardinality(n);
}

} // namespace duckdb
