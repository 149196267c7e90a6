// gg_ingest.cpp — base-table ingest: DuckDB storage -> sink operators -> pinned staging -> HBM.
//
// SURVEY.md §8(f).3.  The reference feeds a hash-join build from PhysicalTableScan: worker threads take
// row groups (122 880-row morsels, src/include/duckdb/storage/table/row_group.hpp:38-39) from a shared
// ParallelTableScanState and call DataTable::Scan per 1024-row vector
// (src/function/table/table_scan.cpp:65-110, src/storage/data_table.cpp:288-345).  The same calls are made
// here, on the reference's own TaskScheduler threads and in the calling query's transaction, with the
// graph sinks at the end of the "pipeline": each task scans a morsel and hands the vectors to
// PhysicalGG*Sink::Sink, whose gg_*_append copies them into the pinned staging block under the staging
// lock; full blocks go to the device asynchronously while the other tasks keep scanning
// (duckdb_pgq_amd/csrc/gg_runtime.hip).  Anything that is not a plain table falls back to a statement on
// a side connection, pulled chunk by chunk.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "duckdb.hpp"
#include "duckdb/catalog/catalog.hpp"
#include "duckdb/catalog/catalog_entry/table_catalog_entry.hpp"
#include "duckdb/common/exception.hpp"
#include "duckdb/main/client_context.hpp"
#include "duckdb/main/connection.hpp"
#include "duckdb/parallel/event.hpp"
#include "duckdb/parallel/pipeline.hpp"
#include "duckdb/parallel/task_scheduler.hpp"
#include "duckdb/parallel/thread_context.hpp"
#include "duckdb/storage/data_table.hpp"
#include "duckdb/transaction/transaction.hpp"
#include "gg_extension.hpp"

namespace duckdb {

namespace {

class GGNoopEvent : public Event {
public:
	explicit GGNoopEvent(Executor &executor) : Event(executor) {
	}
	void Schedule() override {
	}
};

//! What the scan tasks of one table share.
struct IngestState {
	IngestState(ClientContext &context_p, DataTable &storage_p, PhysicalOperator &sink_p)
	    : context(context_p), storage(storage_p), sink(sink_p), transaction(Transaction::GetTransaction(context_p)) {
	}
	ClientContext &context;
	DataTable &storage;
	PhysicalOperator &sink;
	Transaction &transaction;
	vector<column_t> column_ids;
	vector<LogicalType> types;
	ParallelTableScanState parallel_state;
	mutex lock; // guards parallel_state and error
	std::atomic<idx_t> pending {0};
	std::atomic<bool> failed {false};
	string error;
};

class IngestTask : public Task {
public:
	explicit IngestTask(IngestState &state_p) : state(state_p) {
	}
	IngestState &state;

	void Execute() override {
		try {
			Scan();
		} catch (std::exception &ex) {
			lock_guard<mutex> guard(state.lock);
			if (!state.failed) {
				state.error = ex.what();
				state.failed = true;
			}
		}
		state.pending--;
	}

	void Scan() {
		ThreadContext thread(state.context);
		ExecutionContext ec(state.context, thread);
		auto local_sink = state.sink.GetLocalSinkState(ec);
		TableScanState scan;
		// RowGroupScanState's constructor leaves row_group unset (scan_state.hpp:91-97), and NextParallelScan's last
		// task — the transaction-local rows — does not set it either (data_table.cpp:340-349): a task whose FIRST
		// morsel is that one would have DataTable::ScanBaseTable walk a row group at whatever the stack held.  (The
		// reference keeps its scan state in a fresh heap object; here it lives on a worker's stack.  Seen as one
		// fault in a few hundred pinned builds, in RowGroup::InitializeScan below IngestTask::Scan.)
		scan.row_group_scan_state.row_group = nullptr;
		auto column_ids = state.column_ids; // DataTable::Scan wants a mutable vector
		DataChunk chunk;
		chunk.Initialize(state.types);
		while (!state.failed && !state.context.interrupted) {
			{
				lock_guard<mutex> guard(state.lock);
				if (!state.storage.NextParallelScan(state.context, state.parallel_state, scan, column_ids)) {
					break;
				}
			}
			while (true) {
				chunk.Reset();
				state.storage.Scan(state.transaction, chunk, scan, column_ids);
				if (chunk.size() == 0) {
					break;
				}
				state.sink.Sink(ec, *state.sink.sink_state, *local_sink, chunk);
			}
		}
		state.sink.Combine(ec, *state.sink.sink_state, *local_sink);
	}
};

static constexpr idx_t SMALL_TABLE_ROWS = 1u << 20;

void IngestTable(ClientContext &context, const GGScanSource &source, PhysicalOperator &sink) {
	auto &table = *source.table;
	IngestState state(context, *table.storage, sink);
	state.column_ids = source.columns;
	for (auto column : source.columns) {
		state.types.push_back(column == COLUMN_IDENTIFIER_ROW_ID ? LogicalType::BIGINT : table.columns[column].type);
	}
	state.storage.InitializeParallelScan(context, state.parallel_state);

	auto &scheduler = TaskScheduler::GetScheduler(context);
	// the scan ends in a PCIe copy: measured on the MI355X host (PRAGMA threads=256, 40 M rows), 4 tasks
	// stage the table in 29 ms, 8 in 21-28 ms, 16 in 34-50 ms, 32 in 55-70 ms — past eight the tasks only
	// contend for the staging block — so 8 (GG_INGEST_TASKS) is the ceiling whatever PRAGMA threads says
	static const idx_t task_cap = [] {
		auto env = std::getenv("GG_INGEST_TASKS");
		const idx_t n = env ? (idx_t)std::strtoull(env, nullptr, 10) : 8;
		return MaxValue<idx_t>(1, n);
	}();
	idx_t tasks = MaxValue<idx_t>(
	    1, MinValue<idx_t>(MinValue<idx_t>((idx_t)scheduler.NumberOfThreads(), task_cap), state.storage.MaxThreads(context)));
	// a small table (the seeds of a shortest-path plan are read from the vertex table: 0.45 M keys at SF100) is read by
	// the calling thread alone: waking workers for a few row groups costs more than the rows (GG_TIMING: 1.5-2.1 ms with
	// four tasks)
	if (state.storage.GetTotalRows() <= SMALL_TABLE_ROWS) {
		tasks = 1;
	}
	auto producer = scheduler.CreateProducer();
	state.pending = tasks;
	for (idx_t i = 0; i < tasks; i++) {
		scheduler.ScheduleTask(*producer, make_unique<IngestTask>(state));
	}
	// like Executor::WorkOnTasks (src/parallel/executor.cpp): the calling thread works too, then waits for
	// the tasks other threads picked up
	unique_ptr<Task> task;
	while (state.pending > 0) {
		if (scheduler.GetTaskFromProducer(*producer, task)) {
			task->Execute();
			task.reset();
		} else {
			std::this_thread::yield();
		}
	}
	if (state.failed) {
		throw IOException("gg: scanning " + table.name + " failed: " + state.error);
	}
	if (context.interrupted) {
		throw InterruptException();
	}
}

void IngestStatement(ClientContext &context, const string &sql, PhysicalOperator &sink) {
	Connection con(*context.db);
	auto result = con.SendQuery(sql);
	if (!result->success) {
		throw BinderException("gg: scanning the base table failed: " + result->error);
	}
	ThreadContext thread(context);
	ExecutionContext ec(context, thread);
	auto local_sink = sink.GetLocalSinkState(ec);
	while (true) {
		auto chunk = result->Fetch();
		if (!chunk || chunk->size() == 0) {
			break;
		}
		sink.Sink(ec, *sink.sink_state, *local_sink, *chunk);
	}
	sink.Combine(ec, *sink.sink_state, *local_sink);
}

} // namespace

void GGRunSinkPipeline(ClientContext &context, const GGScanSource &source, PhysicalOperator &sink) {
	static const bool timing = std::getenv("GG_TIMING") != nullptr;
	auto t0 = std::chrono::steady_clock::now();
	sink.sink_state = sink.GetGlobalSinkState(context);
	if (source.table) {
		IngestTable(context, source, sink);
	} else {
		IngestStatement(context, source.sql, sink);
	}
	auto t1 = std::chrono::steady_clock::now();
	Pipeline pipeline(context.executor);
	GGNoopEvent event(context.executor);
	sink.Finalize(pipeline, event, context, *sink.sink_state);
	if (timing) {
		auto t2 = std::chrono::steady_clock::now();
		fprintf(stderr, "[gg]   %-20s scan+sink %8.3f ms, finalize %8.3f ms\n", sink.GetName().c_str(),
		        std::chrono::duration<double, std::milli>(t1 - t0).count(),
		        std::chrono::duration<double, std::milli>(t2 - t1).count());
	}
}

namespace {
//! a sink that just collects one key column on the host (for source-vertex lists)
class CollectGlobalState : public GlobalSinkState {
public:
	mutex lock;
	vector<int64_t> values;
};
class CollectLocalState : public LocalSinkState {
public:
	vector<vector<int64_t>> scratch;
	vector<const int64_t *> keys;
};
class PhysicalGGCollect : public PhysicalOperator {
public:
	PhysicalGGCollect() : PhysicalOperator(PhysicalOperatorType::INVALID, {LogicalType::BIGINT}, 0) {
	}
	unique_ptr<GlobalSinkState> GetGlobalSinkState(ClientContext &context) const override {
		return make_unique<CollectGlobalState>();
	}
	unique_ptr<LocalSinkState> GetLocalSinkState(ExecutionContext &context) const override {
		return make_unique<CollectLocalState>();
	}
	SinkResultType Sink(ExecutionContext &context, GlobalSinkState &gstate_p, LocalSinkState &lstate_p,
	                    DataChunk &input) const override {
		auto &gstate = (CollectGlobalState &)gstate_p;
		auto &lstate = (CollectLocalState &)lstate_p;
		const idx_t n = GGKeyColumns(input, {0}, lstate.scratch, lstate.keys);
		lock_guard<mutex> guard(gstate.lock);
		gstate.values.insert(gstate.values.end(), lstate.keys[0], lstate.keys[0] + n);
		return SinkResultType::NEED_MORE_INPUT;
	}
	bool IsSink() const override {
		return true;
	}
	bool ParallelSink() const override {
		return true;
	}
	string GetName() const override {
		return "GG_COLLECT";
	}
};
} // namespace

vector<int64_t> GGScanInt64Column(ClientContext &context, const GGScanSource &source) {
	PhysicalGGCollect collect;
	GGRunSinkPipeline(context, source, collect);
	return move(((CollectGlobalState &)*collect.sink_state).values);
}

GGScanSource GGTableSource(ClientContext &context, const string &table_name, const vector<string> &columns,
                           bool with_rowid) {
	GGScanSource source;
	auto entry = Catalog::GetCatalog(context).GetEntry(context, CatalogType::TABLE_ENTRY, DEFAULT_SCHEMA, table_name,
	                                                   true);
	if (entry && entry->type == CatalogType::TABLE_ENTRY) {
		auto table = (TableCatalogEntry *)entry;
		bool found = true;
		for (auto &name : columns) {
			auto it = table->name_map.find(name);
			if (it == table->name_map.end()) {
				found = false;
				break;
			}
			source.columns.push_back(it->second);
		}
		if (found) {
			if (with_rowid) {
				source.columns.push_back(COLUMN_IDENTIFIER_ROW_ID);
			}
			source.table = table;
			return source;
		}
		source.columns.clear();
	}
	// a view, a schema-qualified name, ...: let the binder sort it out (and report unknown names)
	source.sql = "SELECT ";
	for (idx_t i = 0; i < columns.size(); i++) {
		source.sql += (i ? ", " : "") + GGQuote(columns[i]);
	}
	source.sql += string(with_rowid ? ", rowid" : "") + " FROM " + GGQuote(table_name);
	return source;
}

} // namespace duckdb
