// gg_plan_rule.cpp — planner rules: hash-join chains over an edge table become one GPU path expansion.
//
// SURVEY.md §8(f).1.  The reference plans a k-hop pattern written as SQL
//     FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id         (interactive-complex-3.sql:9-11)
//     FROM person p0, knows k1, person p1, ... WHERE p0.id = k1.src AND k1.dst = p1.id ...
// as a tree of LogicalComparisonJoin over LogicalGet(seq_scan) leaves, which
// PhysicalPlanGenerator::CreatePlan(LogicalComparisonJoin &) turns into PhysicalHashJoin operators
// (src/execution/physical_plan/plan_comparison_join.cpp:146-220).  The rules below look at the same
// logical subtree at the same moment and, when it is exactly a walk pattern, return a PhysicalTableScan
// over the gg scan function (gg_extension.hpp) instead:
//
//   join rule       the subtree's output columns are the walk's vertices -> materialised k-hop expansion
//                   under a projection that restores the join's column layout;
//   aggregate rule  ungrouped count(*) directly over such a subtree -> the count-only expansion (nothing
//                   is materialised), one output row;
//   join rule 2     walks whose vertices all share a neighbour in a second edge table (Train Benchmark
//                   ConnectedSegments, 11 hash joins) -> PhysicalGGFilteredPaths (see PlanSameNeighbourPaths);
//   aggregate rule 2  min(hop) GROUP BY (start, friend) over the `friends` recursive CTE of
//                   bi-10-shortestpath.sql -> the 64-lane bitset BFS (see PlanShortestPath below).
//
// A pattern is accepted only when the substitution is exact for every database state:
//   * every leaf is a plain sequential scan, every join INNER with only column = column conditions;
//     pushed-down filters must be on key columns: `= constant` on the walk's first vertex becomes the single
//     source, comparisons with constants on other positions stay as a filter above the GPU scan;
//   * the edge instances form one path  e1.dst = e2.src, e2.dst = e3.src, ...  and nothing else is
//     equated;
//   * either every walk position is also joined to an instance of ONE vertex table whose key column
//     carries a PRIMARY KEY / UNIQUE constraint (then edges with a dangling endpoint drop out on both
//     sides), or no position is (then the vertex set is the set of endpoint ids, gg_vertices_from_edges,
//     and both edge columns must be declared NOT NULL because the chain's outer ends are not join keys);
//   * the subtree exposes only the key columns (no rowid, no payload columns).
// Anything else is left to the reference's own planner, including sub-chains of a larger join tree: when
// the top join is declined the reference recurses and the rule sees the children again.
//
// How the rules get called: gg_plan_hook.c (interposition shim; INTEGRATION.md §3 shows the equivalent
// three-line patch for a writable tree).  They are inert until `PRAGMA enable_gpu_graph` (or GG_PLAN_RULE=1
// in the environment when the extension is loaded).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <dlfcn.h>
#include <new>

#include "duckdb.hpp"
#include "duckdb/catalog/catalog.hpp"
#include "duckdb/catalog/catalog_entry/schema_catalog_entry.hpp"
#include "duckdb/catalog/catalog_entry/table_catalog_entry.hpp"
#include "duckdb/execution/operator/filter/physical_filter.hpp"
#include "duckdb/execution/operator/projection/physical_projection.hpp"
#include "duckdb/execution/operator/scan/physical_table_scan.hpp"
#include "duckdb/function/pragma_function.hpp"
#include "duckdb/function/table/table_scan.hpp"
#include "duckdb/parser/parsed_data/create_pragma_function_info.hpp"
#include "duckdb/planner/constraints/bound_not_null_constraint.hpp"
#include "duckdb/planner/constraints/bound_unique_constraint.hpp"
#include "duckdb/planner/expression/bound_aggregate_expression.hpp"
#include "duckdb/planner/expression/bound_case_expression.hpp"
#include "duckdb/planner/expression/bound_comparison_expression.hpp"
#include <unordered_set>

#include "duckdb/planner/expression/bound_conjunction_expression.hpp"
#include "duckdb/planner/expression/bound_constant_expression.hpp"
#include "duckdb/planner/expression/bound_function_expression.hpp"
#include "duckdb/planner/expression/bound_operator_expression.hpp"
#include "duckdb/planner/expression/bound_cast_expression.hpp"
#include "duckdb/planner/expression/bound_reference_expression.hpp"
#include "duckdb/planner/filter/conjunction_filter.hpp"
#include "duckdb/planner/filter/constant_filter.hpp"
#include "duckdb/common/types/chunk_collection.hpp"
#include "duckdb/planner/operator/logical_aggregate.hpp"
#include "duckdb/planner/operator/logical_chunk_get.hpp"
#include "duckdb/planner/operator/logical_comparison_join.hpp"
#include "duckdb/planner/operator/logical_cteref.hpp"
#include "duckdb/planner/operator/logical_distinct.hpp"
#include "duckdb/planner/operator/logical_filter.hpp"
#include "duckdb/planner/operator/logical_projection.hpp"
#include "duckdb/planner/operator/logical_recursive_cte.hpp"
#include "duckdb/planner/operator/logical_get.hpp"
#include "duckdb/planner/operator/logical_insert.hpp"
#include "duckdb/planner/operator/logical_delete.hpp"
#include "duckdb/transaction/transaction.hpp"
#include "duckdb/planner/operator/logical_update.hpp"
#include "gg_extension.hpp"
#include "gg_pipeline.hpp"
#include "gg_plan_hook.h"
// The rule needs the ClientContext a plan is made for (the switches are per connection), and the generator keeps
// it private.  A reference with oracle/callout.patch hands it to the call-out (PlanCallouts::plan_fn), so the build
// for that reference (-DGG_REFERENCE_CALLOUTS) reads the class as it is.  Behind the interposition shim — a stock
// reference — there is nobody to hand it over: the header is included last, so that only this one class definition
// is read with the access specifier widened; the layout does not change.
#ifdef GG_REFERENCE_CALLOUTS
#include "duckdb/execution/physical_plan_generator.hpp"
#else
#define private public
#include "duckdb/execution/physical_plan_generator.hpp"
#undef private
#endif

namespace duckdb {

static std::atomic<uint64_t> g_rules_fired {0};
static bool g_callouts_registered = false; // rules registered with a patched reference's call-outs (no shim)

namespace {

//! One base-table leaf of the join tree.
struct ScanLeaf {
	LogicalGet *get;
	TableCatalogEntry *table;
};

//! A column of a leaf: (index into PatternInput::leaves, column id inside the table)
struct LeafColumn {
	idx_t leaf;
	column_t column;
	bool operator==(const LeafColumn &o) const {
		return leaf == o.leaf && column == o.column;
	}
};

//! `column = constant` pushed into a leaf's scan
struct LeafConstant {
	LeafColumn column;
	int64_t value;
	TableFilter *filter;
};

//! any other filter pushed into a leaf's scan (<>, <, >=, ranges, ORs of those) on an integer column
struct LeafFilter {
	LeafColumn column;
	TableFilter *filter;
};

struct PatternInput {
	vector<ScanLeaf> leaves;
	vector<std::pair<LeafColumn, LeafColumn>> equalities;
	vector<LeafConstant> constants;
	vector<LeafFilter> filters;
	//! filters pushed into a leaf's scan on a column that is no integer key (VARCHAR, DATE, ...): only ever a predicate
	//! on a PAYLOAD column of an edge instance (join rule 3 evaluates it above the rows it fetches by rowid)
	vector<LeafFilter> other_filters;
	vector<unique_ptr<TableFilter>> owned_filters; // filters rebuilt from a LogicalFilter above a leaf
	//! pure column projections above a leaf (column pruning leaves one after a filter whose column is not
	//! needed further up): projection table index -> the leaf column behind each of its outputs
	vector<std::pair<idx_t, vector<LeafColumn>>> aliases;
};

//! What the pattern turned out to be.
struct WalkPattern {
	TableCatalogEntry *edge_table = nullptr;
	column_t src_column = 0, dst_column = 0;
	TableCatalogEntry *vertex_table = nullptr; // null: vertex set = endpoint ids
	column_t vertex_key = 0;
	idx_t hops = 0;
	bool all_sources = true;
	vector<int64_t> sources; // walk position 0 is pinned to a constant (k1.src = C, interactive-complex-3.sql:9)
	//! predicates on other walk positions (k2.dst <> X, interactive-complex-3.sql:11): kept as a filter
	//! above the GPU scan.  (walk position, the pushed-down filter) — the filter outlives planning inside
	//! the logical operator tree only, so it is turned into an expression before the rule returns
	vector<std::pair<idx_t, TableFilter *>> residual;
	//! predicates on payload columns of edge instances (k2.creationDate > D, k1.weight = 3): (1-based edge number,
	//! column of the edge table, the pushed-down filter) — evaluated on the columns PhysicalGGPathEdges fetches by rowid
	struct PayloadFilter {
		idx_t edge;
		column_t column;
		TableFilter *filter;
	};
	vector<PayloadFilter> payload_filters;
	//! per leaf: walk position of its src column (edge leaves: position of dst is +1) or of its key
	vector<idx_t> edge_position;   // leaf -> 1-based edge number, 0 for vertex leaves
	vector<idx_t> vertex_position; // leaf -> walk position (vertex leaves only)
};

bool ResolveLeafColumn(const PatternInput &in, const ColumnBinding &binding, LeafColumn &out) {
	for (auto &alias : in.aliases) {
		if (alias.first == binding.table_index) {
			if (binding.column_index >= alias.second.size()) {
				return false;
			}
			out = alias.second[binding.column_index];
			return true;
		}
	}
	for (idx_t l = 0; l < in.leaves.size(); l++) {
		auto &get = *in.leaves[l].get;
		if (get.table_index != binding.table_index) {
			continue;
		}
		if (binding.column_index >= get.column_ids.size()) {
			return false;
		}
		const auto column = get.column_ids[binding.column_index];
		if (column == COLUMN_IDENTIFIER_ROW_ID || column >= in.leaves[l].table->columns.size()) {
			return false;
		}
		out = {l, column};
		return true;
	}
	return false;
}

bool ColumnIsIntegerKey(TableCatalogEntry &table, column_t column) {
	const auto id = table.columns[column].type.id();
	return id == LogicalTypeId::BIGINT || id == LogicalTypeId::INTEGER;
}

//! filter == (column = value), possibly ANDed with IS NOT NULL
bool IsEqualityWithConstant(TableFilter &filter, int64_t &value) {
	switch (filter.filter_type) {
	case TableFilterType::CONSTANT_COMPARISON: {
		auto &constant = (ConstantFilter &)filter;
		if (constant.comparison_type != ExpressionType::COMPARE_EQUAL || constant.constant.is_null ||
		    !constant.constant.type().IsIntegral()) {
			return false;
		}
		value = constant.constant.GetValue<int64_t>();
		return true;
	}
	case TableFilterType::CONJUNCTION_AND: {
		bool found = false;
		for (auto &child : ((ConjunctionAndFilter &)filter).child_filters) {
			if (child->filter_type == TableFilterType::IS_NOT_NULL) {
				continue;
			}
			int64_t v;
			if (!IsEqualityWithConstant(*child, v) || (found && v != value)) {
				return false;
			}
			value = v;
			found = true;
		}
		return found;
	}
	default:
		return false;
	}
}

//! A pushed-down filter over an integer column as SQL text; false if the filter is of an unknown kind.
//! The same filters as a host-side predicate over int64 keys: the seeds of a shortest-path plan are read from the
//! vertex table in the statement's OWN transaction (a side connection would see another snapshot) and filtered here.
struct KeyPredicate {
	enum Kind { IS_NOT_NULL, COMPARE, AND, OR } kind = IS_NOT_NULL;
	ExpressionType comparison = ExpressionType::COMPARE_EQUAL;
	int64_t constant = 0;
	vector<KeyPredicate> children;
	bool Accepts(int64_t v) const {
		switch (kind) {
		case IS_NOT_NULL:
			return true; // NULL keys never reach here
		case COMPARE:
			switch (comparison) {
			case ExpressionType::COMPARE_EQUAL:
				return v == constant;
			case ExpressionType::COMPARE_NOTEQUAL:
				return v != constant;
			case ExpressionType::COMPARE_LESSTHAN:
				return v < constant;
			case ExpressionType::COMPARE_LESSTHANOREQUALTO:
				return v <= constant;
			case ExpressionType::COMPARE_GREATERTHAN:
				return v > constant;
			default:
				return v >= constant;
			}
		case AND:
			for (auto &c : children) {
				if (!c.Accepts(v)) {
					return false;
				}
			}
			return true;
		default:
			for (auto &c : children) {
				if (c.Accepts(v)) {
					return true;
				}
			}
			return false;
		}
	}
};

bool FilterToPredicate(TableFilter &filter, KeyPredicate &out) {
	switch (filter.filter_type) {
	case TableFilterType::IS_NOT_NULL:
		out.kind = KeyPredicate::IS_NOT_NULL;
		return true;
	case TableFilterType::CONSTANT_COMPARISON: {
		auto &constant = (ConstantFilter &)filter;
		if (constant.constant.is_null || !constant.constant.type().IsIntegral()) {
			return false;
		}
		switch (constant.comparison_type) {
		case ExpressionType::COMPARE_EQUAL:
		case ExpressionType::COMPARE_NOTEQUAL:
		case ExpressionType::COMPARE_LESSTHAN:
		case ExpressionType::COMPARE_LESSTHANOREQUALTO:
		case ExpressionType::COMPARE_GREATERTHAN:
		case ExpressionType::COMPARE_GREATERTHANOREQUALTO:
			break;
		default:
			return false;
		}
		out.kind = KeyPredicate::COMPARE;
		out.comparison = constant.comparison_type;
		out.constant = constant.constant.GetValue<int64_t>();
		return true;
	}
	case TableFilterType::CONJUNCTION_AND:
	case TableFilterType::CONJUNCTION_OR: {
		const bool is_and = filter.filter_type == TableFilterType::CONJUNCTION_AND;
		auto &children = is_and ? ((ConjunctionAndFilter &)filter).child_filters
		                        : ((ConjunctionOrFilter &)filter).child_filters;
		if (children.empty()) {
			return false;
		}
		out.kind = is_and ? KeyPredicate::AND : KeyPredicate::OR;
		for (auto &child : children) {
			KeyPredicate c;
			if (!FilterToPredicate(*child, c)) {
				return false;
			}
			out.children.push_back(move(c));
		}
		return true;
	}
	default:
		return false;
	}
}

bool FilterToSQL(TableFilter &filter, const string &column, string &out) {
	switch (filter.filter_type) {
	case TableFilterType::IS_NOT_NULL:
		out = column + " IS NOT NULL";
		return true;
	case TableFilterType::CONSTANT_COMPARISON: {
		auto &constant = (ConstantFilter &)filter;
		if (constant.constant.is_null || !constant.constant.type().IsIntegral()) {
			return false;
		}
		const char *op;
		switch (constant.comparison_type) {
		case ExpressionType::COMPARE_EQUAL:
			op = " = ";
			break;
		case ExpressionType::COMPARE_NOTEQUAL:
			op = " <> ";
			break;
		case ExpressionType::COMPARE_LESSTHAN:
			op = " < ";
			break;
		case ExpressionType::COMPARE_LESSTHANOREQUALTO:
			op = " <= ";
			break;
		case ExpressionType::COMPARE_GREATERTHAN:
			op = " > ";
			break;
		case ExpressionType::COMPARE_GREATERTHANOREQUALTO:
			op = " >= ";
			break;
		default:
			return false;
		}
		out = column + op + to_string(constant.constant.GetValue<int64_t>());
		return true;
	}
	case TableFilterType::CONJUNCTION_AND:
	case TableFilterType::CONJUNCTION_OR: {
		const bool is_and = filter.filter_type == TableFilterType::CONJUNCTION_AND;
		auto &children = is_and ? ((ConjunctionAndFilter &)filter).child_filters
		                        : ((ConjunctionOrFilter &)filter).child_filters;
		if (children.empty()) {
			return false;
		}
		out = "(";
		for (idx_t i = 0; i < children.size(); i++) {
			string child;
			if (!FilterToSQL(*children[i], column, child)) {
				return false;
			}
			out += (i ? (is_and ? " AND " : " OR ") : "") + child;
		}
		out += ")";
		return true;
	}
	default:
		return false;
	}
}

//! The same filter as an expression over scan column `column` (BIGINT); shapes FilterToSQL accepts.
unique_ptr<Expression> FilterToExpression(TableFilter &filter, idx_t column) {
	switch (filter.filter_type) {
	case TableFilterType::IS_NOT_NULL: {
		auto result = make_unique<BoundOperatorExpression>(ExpressionType::OPERATOR_IS_NOT_NULL, LogicalType::BOOLEAN);
		result->children.push_back(make_unique<BoundReferenceExpression>(LogicalType::BIGINT, column));
		return move(result);
	}
	case TableFilterType::CONSTANT_COMPARISON: {
		auto &constant = (ConstantFilter &)filter;
		return make_unique<BoundComparisonExpression>(
		    constant.comparison_type, make_unique<BoundReferenceExpression>(LogicalType::BIGINT, column),
		    make_unique<BoundConstantExpression>(Value::BIGINT(constant.constant.GetValue<int64_t>())));
	}
	case TableFilterType::CONJUNCTION_AND:
	case TableFilterType::CONJUNCTION_OR: {
		const bool is_and = filter.filter_type == TableFilterType::CONJUNCTION_AND;
		auto &children = is_and ? ((ConjunctionAndFilter &)filter).child_filters
		                        : ((ConjunctionOrFilter &)filter).child_filters;
		auto result = make_unique<BoundConjunctionExpression>(is_and ? ExpressionType::CONJUNCTION_AND
		                                                             : ExpressionType::CONJUNCTION_OR);
		for (auto &child : children) {
			result->children.push_back(FilterToExpression(*child, column));
		}
		return move(result);
	}
	default:
		throw InternalException("gg: unexpected table filter");
	}
}

//! A predicate the optimizer left in a LogicalFilter above a scan (it pushes =, <, >, ranges into the scan
//! but not <>), restated as a table filter on ONE column: comparisons of a column reference with an integer
//! constant, and AND/OR of those over the same column.  ref_index receives the referenced child column.
unique_ptr<TableFilter> ExpressionToFilter(Expression &expr, idx_t &ref_index) {
	switch (expr.type) {
	case ExpressionType::COMPARE_EQUAL:
	case ExpressionType::COMPARE_NOTEQUAL:
	case ExpressionType::COMPARE_LESSTHAN:
	case ExpressionType::COMPARE_LESSTHANOREQUALTO:
	case ExpressionType::COMPARE_GREATERTHAN:
	case ExpressionType::COMPARE_GREATERTHANOREQUALTO: {
		auto &cmp = (BoundComparisonExpression &)expr;
		auto type = expr.type;
		Expression *ref = cmp.left.get(), *constant = cmp.right.get();
		if (ref->type != ExpressionType::BOUND_REF) { // constant on the left: mirror the comparison
			std::swap(ref, constant);
			switch (type) {
			case ExpressionType::COMPARE_LESSTHAN:
				type = ExpressionType::COMPARE_GREATERTHAN;
				break;
			case ExpressionType::COMPARE_LESSTHANOREQUALTO:
				type = ExpressionType::COMPARE_GREATERTHANOREQUALTO;
				break;
			case ExpressionType::COMPARE_GREATERTHAN:
				type = ExpressionType::COMPARE_LESSTHAN;
				break;
			case ExpressionType::COMPARE_GREATERTHANOREQUALTO:
				type = ExpressionType::COMPARE_LESSTHANOREQUALTO;
				break;
			default:
				break;
			}
		}
		if (ref->type != ExpressionType::BOUND_REF || constant->type != ExpressionType::VALUE_CONSTANT) {
			return nullptr;
		}
		auto &value = ((BoundConstantExpression &)*constant).value;
		const auto index = ((BoundReferenceExpression &)*ref).index;
		if (value.is_null || !value.type().IsIntegral() || (ref_index != INVALID_INDEX && ref_index != index)) {
			return nullptr;
		}
		ref_index = index;
		return make_unique<ConstantFilter>(type, Value::BIGINT(value.GetValue<int64_t>()));
	}
	case ExpressionType::CONJUNCTION_AND:
	case ExpressionType::CONJUNCTION_OR: {
		auto &conjunction = (BoundConjunctionExpression &)expr;
		vector<unique_ptr<TableFilter>> children;
		for (auto &child : conjunction.children) {
			auto filter = ExpressionToFilter(*child, ref_index);
			if (!filter) {
				return nullptr;
			}
			children.push_back(move(filter));
		}
		if (expr.type == ExpressionType::CONJUNCTION_AND) {
			auto result = make_unique<ConjunctionAndFilter>();
			result->child_filters = move(children);
			return move(result);
		}
		auto result = make_unique<ConjunctionOrFilter>();
		result->child_filters = move(children);
		return move(result);
	}
	default:
		return nullptr;
	}
}

//! A pushed-down filter on a column of ANY type, of the shapes FilterCombiner emits (filter_combiner.cpp): comparisons
//! with a constant, IS [NOT] NULL, AND / OR of those.
bool PayloadFilterSupported(TableFilter &filter) {
	switch (filter.filter_type) {
	case TableFilterType::IS_NULL:
	case TableFilterType::IS_NOT_NULL:
	case TableFilterType::CONSTANT_COMPARISON:
		return true;
	case TableFilterType::CONJUNCTION_AND:
		for (auto &child : ((ConjunctionAndFilter &)filter).child_filters) {
			if (!PayloadFilterSupported(*child)) {
				return false;
			}
		}
		return true;
	case TableFilterType::CONJUNCTION_OR:
		for (auto &child : ((ConjunctionOrFilter &)filter).child_filters) {
			if (!PayloadFilterSupported(*child)) {
				return false;
			}
		}
		return true;
	default:
		return false;
	}
}

//! The same filter as an expression over scan column `column` of type `type`
unique_ptr<Expression> PayloadFilterToExpression(TableFilter &filter, idx_t column, const LogicalType &type) {
	switch (filter.filter_type) {
	case TableFilterType::IS_NULL:
	case TableFilterType::IS_NOT_NULL: {
		auto result = make_unique<BoundOperatorExpression>(filter.filter_type == TableFilterType::IS_NULL
		                                                       ? ExpressionType::OPERATOR_IS_NULL
		                                                       : ExpressionType::OPERATOR_IS_NOT_NULL,
		                                                   LogicalType::BOOLEAN);
		result->children.push_back(make_unique<BoundReferenceExpression>(type, column));
		return move(result);
	}
	case TableFilterType::CONSTANT_COMPARISON: {
		auto &constant = (ConstantFilter &)filter;
		return make_unique<BoundComparisonExpression>(constant.comparison_type,
		                                              make_unique<BoundReferenceExpression>(type, column),
		                                              make_unique<BoundConstantExpression>(constant.constant.CastAs(type)));
	}
	case TableFilterType::CONJUNCTION_AND:
	case TableFilterType::CONJUNCTION_OR: {
		const bool is_and = filter.filter_type == TableFilterType::CONJUNCTION_AND;
		auto &children = is_and ? ((ConjunctionAndFilter &)filter).child_filters
		                        : ((ConjunctionOrFilter &)filter).child_filters;
		auto result = make_unique<BoundConjunctionExpression>(is_and ? ExpressionType::CONJUNCTION_AND
		                                                             : ExpressionType::CONJUNCTION_OR);
		for (auto &child : children) {
			result->children.push_back(PayloadFilterToExpression(*child, column, type));
		}
		return move(result);
	}
	default:
		throw InternalException("gg: unexpected table filter");
	}
}

//! Flatten a tree of inner equi-joins over sequential scans; false if anything else is in it.
bool CollectJoinTree(LogicalOperator &op, PatternInput &in) {
	switch (op.type) {
	case LogicalOperatorType::LOGICAL_GET: {
		auto &get = (LogicalGet &)op;
		if (!get.children.empty() || get.function.name != "seq_scan" || !get.bind_data) {
			return false;
		}
		auto &bind = (TableScanBindData &)*get.bind_data;
		if (bind.is_index_scan || !bind.table) {
			return false;
		}
		in.leaves.push_back({&get, bind.table});
		// pushed-down filters: only `column = integer constant` (FilterCombiner emits it as
		// ConstantFilter AND IsNotNullFilter, src/optimizer/filter_combiner.cpp:473-475)
		for (auto &entry : get.table_filters.filters) {
			int64_t value;
			string ignored;
			if (entry.first >= bind.table->columns.size()) {
				return false;
			}
			if (!ColumnIsIntegerKey(*bind.table, entry.first)) {
				if (!PayloadFilterSupported(*entry.second)) {
					return false;
				}
				in.other_filters.push_back({{in.leaves.size() - 1, entry.first}, entry.second.get()});
				continue;
			}
			if (IsEqualityWithConstant(*entry.second, value)) {
				in.constants.push_back({{in.leaves.size() - 1, entry.first}, value, entry.second.get()});
			} else if (FilterToSQL(*entry.second, "c", ignored)) { // a shape FilterToExpression understands
				in.filters.push_back({{in.leaves.size() - 1, entry.first}, entry.second.get()});
			} else {
				return false;
			}
		}
		return true;
	}
	case LogicalOperatorType::LOGICAL_PROJECTION: {
		// a projection that only passes columns of the scan below it through
		auto &projection = (LogicalProjection &)op;
		if (projection.children.size() != 1 ||
		    (projection.children[0]->type != LogicalOperatorType::LOGICAL_GET &&
		     projection.children[0]->type != LogicalOperatorType::LOGICAL_FILTER) ||
		    !CollectJoinTree(*projection.children[0], in)) {
			return false;
		}
		auto child_bindings = projection.children[0]->GetColumnBindings();
		vector<LeafColumn> columns;
		for (auto &expr : projection.expressions) {
			idx_t index;
			LeafColumn column;
			if (expr->type != ExpressionType::BOUND_REF ||
			    (index = ((BoundReferenceExpression &)*expr).index) >= child_bindings.size() ||
			    !ResolveLeafColumn(in, child_bindings[index], column)) {
				return false;
			}
			columns.push_back(column);
		}
		in.aliases.emplace_back(projection.table_index, move(columns));
		return true;
	}
	case LogicalOperatorType::LOGICAL_FILTER: {
		// predicates the optimizer kept above a scan (e.g. `k2.k_person2id <> X`, interactive-complex-3.sql:11)
		auto &filter = (LogicalFilter &)op;
		// (a projection_map only drops columns from the filter's output — the bindings the joins above see
		// still name the scan's columns; the predicates themselves index the scan's full output)
		if (filter.children.size() != 1 || filter.children[0]->type != LogicalOperatorType::LOGICAL_GET ||
		    !CollectJoinTree(*filter.children[0], in)) {
			return false;
		}
		const idx_t leaf = in.leaves.size() - 1;
		auto &get = *in.leaves[leaf].get;
		for (auto &expr : filter.expressions) {
			idx_t ref_index = INVALID_INDEX;
			auto restated = ExpressionToFilter(*expr, ref_index);
			if (!restated || ref_index >= get.column_ids.size() || get.column_ids[ref_index] == COLUMN_IDENTIFIER_ROW_ID ||
			    !ColumnIsIntegerKey(*in.leaves[leaf].table, get.column_ids[ref_index])) {
				return false;
			}
			in.filters.push_back({{leaf, get.column_ids[ref_index]}, restated.get()});
			in.owned_filters.push_back(move(restated));
		}
		return true;
	}
	case LogicalOperatorType::LOGICAL_COMPARISON_JOIN: {
		auto &join = (LogicalComparisonJoin &)op;
		if (join.join_type != JoinType::INNER || join.children.size() != 2 || join.conditions.empty()) {
			return false;
		}
		if (!CollectJoinTree(*join.children[0], in) || !CollectJoinTree(*join.children[1], in)) {
			return false;
		}
		// conditions are already BoundReferenceExpressions into the children's column lists
		// (ColumnBindingResolver ran in PhysicalPlanGenerator::CreatePlan(unique_ptr<LogicalOperator>),
		// src/execution/physical_plan_generator.cpp:29-33); the logical bindings are still derivable
		auto left_bindings = join.children[0]->GetColumnBindings();
		auto right_bindings = join.children[1]->GetColumnBindings();
		for (auto &cond : join.conditions) {
			if (cond.comparison != ExpressionType::COMPARE_EQUAL || cond.null_values_are_equal ||
			    cond.left->type != ExpressionType::BOUND_REF || cond.right->type != ExpressionType::BOUND_REF) {
				return false;
			}
			const auto li = ((BoundReferenceExpression &)*cond.left).index;
			const auto ri = ((BoundReferenceExpression &)*cond.right).index;
			LeafColumn l, r;
			if (li >= left_bindings.size() || ri >= right_bindings.size() ||
			    !ResolveLeafColumn(in, left_bindings[li], l) || !ResolveLeafColumn(in, right_bindings[ri], r)) {
				return false;
			}
			in.equalities.emplace_back(l, r);
		}
		return true;
	}
	default:
		return false;
	}
}

bool ColumnIsNotNull(TableCatalogEntry &table, column_t column) {
	for (auto &constraint : table.bound_constraints) {
		if (constraint->type == ConstraintType::NOT_NULL && ((BoundNotNullConstraint &)*constraint).index == column) {
			return true;
		}
		if (constraint->type == ConstraintType::UNIQUE) {
			auto &unique = (BoundUniqueConstraint &)*constraint;
			if (unique.is_primary_key && unique.key_set.count(column)) {
				return true;
			}
		}
	}
	return false;
}

bool ColumnIsUnique(TableCatalogEntry &table, column_t column) {
	for (auto &constraint : table.bound_constraints) {
		if (constraint->type == ConstraintType::UNIQUE) {
			auto &unique = (BoundUniqueConstraint &)*constraint;
			if (unique.keys.size() == 1 && unique.keys[0] == column) {
				return true;
			}
		}
	}
	return false;
}

//! Union-find over the leaf columns that appear in equalities.
struct ColumnClasses {
	vector<LeafColumn> members;
	vector<idx_t> parent;

	idx_t Add(const LeafColumn &c) {
		for (idx_t i = 0; i < members.size(); i++) {
			if (members[i] == c) {
				return i;
			}
		}
		members.push_back(c);
		parent.push_back(members.size() - 1);
		return members.size() - 1;
	}
	idx_t Find(idx_t i) {
		while (parent[i] != i) {
			parent[i] = parent[parent[i]];
			i = parent[i];
		}
		return i;
	}
	void Union(idx_t a, idx_t b) {
		parent[Find(a)] = Find(b);
	}
	//! class id of a column, or INVALID_INDEX when the column is in no equality
	idx_t ClassOf(const LeafColumn &c) {
		for (idx_t i = 0; i < members.size(); i++) {
			if (members[i] == c) {
				return Find(i);
			}
		}
		return INVALID_INDEX;
	}
};

//! Try to read the flattened join tree as a walk with the given roles.
bool SolveWithRoles(PatternInput &in, ColumnClasses &classes, TableCatalogEntry *edge_table,
                    TableCatalogEntry *vertex_table, column_t src, column_t dst, WalkPattern &out) {
	const idx_t n = in.leaves.size();
	vector<idx_t> edge_leaves, vertex_leaves;
	for (idx_t l = 0; l < n; l++) {
		if (in.leaves[l].table == edge_table) {
			edge_leaves.push_back(l);
		} else if (in.leaves[l].table == vertex_table) {
			vertex_leaves.push_back(l);
		} else {
			return false;
		}
	}
	const idx_t hops = edge_leaves.size();
	if (hops < 1 || hops > GG_MAX_HOPS) {
		return false;
	}
	// every column in an equality must be an edge endpoint column or the vertex key
	column_t vertex_key = INVALID_INDEX;
	for (auto &member : classes.members) {
		if (in.leaves[member.leaf].table == edge_table) {
			if (member.column != src && member.column != dst) {
				return false;
			}
		} else {
			if (vertex_key == INVALID_INDEX) {
				vertex_key = member.column;
			} else if (vertex_key != member.column) {
				return false;
			}
		}
	}
	// successor relation: e -> f when e.dst and f.src are in one class; every class may hold at most one
	// dst column and one src column of edge leaves
	vector<idx_t> src_class(n, INVALID_INDEX), dst_class(n, INVALID_INDEX);
	for (auto l : edge_leaves) {
		src_class[l] = classes.ClassOf({l, src});
		dst_class[l] = classes.ClassOf({l, dst});
	}
	for (auto a : edge_leaves) {
		for (auto b : edge_leaves) {
			if (a == b) {
				if (src_class[a] != INVALID_INDEX && src_class[a] == dst_class[a]) {
					return false; // e.src = e.dst: a self-loop filter, not a walk
				}
				continue;
			}
			if (src_class[a] != INVALID_INDEX && src_class[a] == src_class[b]) {
				return false;
			}
			if (dst_class[a] != INVALID_INDEX && dst_class[a] == dst_class[b]) {
				return false;
			}
		}
	}
	vector<idx_t> successor(n, INVALID_INDEX), predecessor(n, INVALID_INDEX);
	for (auto a : edge_leaves) {
		for (auto b : edge_leaves) {
			if (a != b && dst_class[a] != INVALID_INDEX && dst_class[a] == src_class[b]) {
				successor[a] = b;
				predecessor[b] = a;
			}
		}
	}
	idx_t first = INVALID_INDEX;
	for (auto l : edge_leaves) {
		if (predecessor[l] == INVALID_INDEX) {
			if (first != INVALID_INDEX) {
				return false; // two chains: the join tree would be a cross product of walks
			}
			first = l;
		}
	}
	if (first == INVALID_INDEX) {
		return false; // a cycle
	}
	out.edge_position.assign(n, 0);
	out.vertex_position.assign(n, INVALID_INDEX);
	vector<idx_t> position_class(hops + 1, INVALID_INDEX);
	idx_t visited = 0;
	for (idx_t l = first; l != INVALID_INDEX; l = successor[l]) {
		if (out.edge_position[l] != 0 || visited == hops) {
			return false;
		}
		out.edge_position[l] = ++visited;
		position_class[visited - 1] = src_class[l];
		position_class[visited] = dst_class[l];
	}
	if (visited != hops) {
		return false;
	}
	// vertex leaves: each one keyed into exactly one walk position, each position at most once
	vector<bool> covered(hops + 1, false);
	for (auto l : vertex_leaves) {
		const auto cls = classes.ClassOf({l, vertex_key});
		if (cls == INVALID_INDEX) {
			return false;
		}
		idx_t pos = INVALID_INDEX;
		for (idx_t p = 0; p <= hops; p++) {
			if (position_class[p] == cls) {
				pos = p;
			}
		}
		if (pos == INVALID_INDEX || covered[pos]) {
			return false;
		}
		covered[pos] = true;
		out.vertex_position[l] = pos;
	}
	// every equality class must be a walk position (nothing else is equated)
	for (idx_t i = 0; i < classes.members.size(); i++) {
		const auto cls = classes.Find(i);
		bool found = false;
		for (idx_t p = 0; p <= hops; p++) {
			found = found || position_class[p] == cls;
		}
		if (!found) {
			return false;
		}
	}
	if (!vertex_leaves.empty()) {
		for (idx_t p = 0; p <= hops; p++) {
			if (!covered[p]) {
				return false; // partly validated walks are not what the GPU operators compute
			}
		}
		if (!ColumnIsUnique(*vertex_table, vertex_key) || !ColumnIsIntegerKey(*vertex_table, vertex_key)) {
			return false;
		}
	} else {
		if (hops < 2 || !ColumnIsNotNull(*edge_table, src) || !ColumnIsNotNull(*edge_table, dst)) {
			return false;
		}
	}
	if (!ColumnIsIntegerKey(*edge_table, src) || !ColumnIsIntegerKey(*edge_table, dst)) {
		return false;
	}
	// predicates pushed into the scans.  A column is a walk position: src of edge e is position e-1, dst is e,
	// a vertex leaf's key its own position.  `= c` on position 0 pins the source (the optimizer copies a
	// constant to every column it is transitively equal to: all copies must agree); everything else stays
	// a filter on the walks.
	auto position_of = [&](const LeafColumn &column, idx_t &position) {
		const auto l = column.leaf;
		if (out.edge_position[l]) {
			if (column.column == src) {
				position = out.edge_position[l] - 1;
			} else if (column.column == dst) {
				position = out.edge_position[l];
			} else {
				return false;
			}
			return true;
		}
		if (out.vertex_position[l] != INVALID_INDEX && column.column == vertex_key) {
			position = out.vertex_position[l];
			return true;
		}
		return false;
	};
	out.all_sources = true;
	out.sources.clear();
	out.residual.clear();
	out.payload_filters.clear();
	// a filter on a column of an EDGE instance that is neither of its keys is a predicate on that edge's payload
	auto payload_of = [&](const LeafColumn &column, TableFilter *filter) {
		const auto l = column.leaf;
		if (!out.edge_position[l] || column.column == src || column.column == dst) {
			return false;
		}
		out.payload_filters.push_back({out.edge_position[l], column.column, filter});
		return true;
	};
	for (auto &filter : in.other_filters) {
		if (!payload_of(filter.column, filter.filter)) {
			return false;
		}
	}
	for (auto &constant : in.constants) {
		idx_t position;
		if (!position_of(constant.column, position)) {
			if (payload_of(constant.column, constant.filter)) {
				continue;
			}
			return false;
		}
		if (position == 0) {
			if (!out.sources.empty() && out.sources[0] != constant.value) {
				return false; // contradictory: the reference returns nothing, leave it to it
			}
			out.sources.assign(1, constant.value);
			out.all_sources = false;
		} else {
			out.residual.emplace_back(position, constant.filter); // a pinned middle or far end: filtered afterwards
		}
	}
	for (auto &filter : in.filters) {
		idx_t position;
		if (!position_of(filter.column, position)) {
			if (payload_of(filter.column, filter.filter)) {
				continue;
			}
			return false;
		}
		out.residual.emplace_back(position, filter.filter);
	}
	out.edge_table = edge_table;
	out.src_column = src;
	out.dst_column = dst;
	out.vertex_table = vertex_leaves.empty() ? nullptr : vertex_table;
	out.vertex_key = vertex_key;
	out.hops = hops;
	return true;
}

bool SolveWalkPattern(PatternInput &in, WalkPattern &out) {
	if (in.leaves.size() < 2 || in.equalities.empty()) {
		return false;
	}
	vector<TableCatalogEntry *> tables;
	for (auto &leaf : in.leaves) {
		if (std::find(tables.begin(), tables.end(), leaf.table) == tables.end()) {
			tables.push_back(leaf.table);
		}
	}
	if (tables.size() > 2) {
		return false;
	}
	ColumnClasses classes;
	for (auto &eq : in.equalities) {
		const auto a = classes.Add(eq.first), b = classes.Add(eq.second);
		classes.Union(a, b);
	}
	for (idx_t t = 0; t < tables.size(); t++) {
		auto edge_table = tables[t];
		auto vertex_table = tables.size() == 2 ? tables[1 - t] : nullptr;
		// the two endpoint columns are the ones the edge leaves use in equalities
		vector<column_t> used;
		for (auto &member : classes.members) {
			if (in.leaves[member.leaf].table == edge_table &&
			    std::find(used.begin(), used.end(), member.column) == used.end()) {
				used.push_back(member.column);
			}
		}
		if (used.size() != 2) {
			continue;
		}
		std::sort(used.begin(), used.end()); // prefer reading the table's first key column as the source
		// both readings of the walk's direction may fit; the one that turns a pinned end into the source
		// (fewer predicates left to filter afterwards) wins
		WalkPattern forward, backward;
		const bool fits_forward = SolveWithRoles(in, classes, edge_table, vertex_table, used[0], used[1], forward);
		const bool fits_backward = SolveWithRoles(in, classes, edge_table, vertex_table, used[1], used[0], backward);
		if (fits_forward && (!fits_backward || forward.residual.size() <= backward.residual.size())) {
			out = forward;
			return true;
		}
		if (fits_backward) {
			out = backward;
			return true;
		}
	}
	return false;
}

//! tables the plan being substituted reads; registered with the generator like LogicalGet's dependency
//! callback does (plan_get.cpp:50-52), so a prepared statement notices when one of them is dropped
thread_local vector<CatalogEntry *> g_plan_tables;
//! the connection whose statement is being planned (set by RuleEntry)
thread_local ClientContext *g_plan_context = nullptr;
//! its plan generator: the key-join rule plans the probe side with it (CreatePlan(unique_ptr<LogicalOperator>), public)
static thread_local PhysicalPlanGenerator *g_plan_generator = nullptr;

GGScanSource TableColumns(TableCatalogEntry *table, vector<column_t> columns) {
	g_plan_tables.push_back(table);
	GGScanSource source;
	source.table = table;
	source.columns = move(columns);
	return source;
}

GGGraphSpec GraphSpecOf(const WalkPattern &pattern) {
	GGGraphSpec spec;
	if (pattern.vertex_table) {
		spec.vertices = TableColumns(pattern.vertex_table, {pattern.vertex_key});
	}
	spec.edges = TableColumns(pattern.edge_table, {pattern.src_column, pattern.dst_column});
	return spec;
}

//! PhysicalTableScan over the gg scan function for `hops`-hop walks of the pattern.
unique_ptr<PhysicalOperator> MakeExpandScan(const WalkPattern &pattern, bool count_only, idx_t estimated_cardinality) {
	auto spec = GraphSpecOf(pattern);
	const int hops = (int)pattern.hops;
	const auto sources = pattern.sources;
	const bool all_sources = pattern.all_sources;
	if (all_sources && hops == 2) {
		// the plan shape whose result adds (counts) or concatenates (rows) over ownership shards of the graph: with
		// GG_DEVICES=N the tables go to N device contexts (device p mod the devices present), N CSR shards are built
		// side by side, and every shard counts or materialises the walks whose middle vertex it owns — bench.py's N
		// ranks inside one process, for a host with several GPUs; the rows of each shard cross its own PCIe link
		spec.shards = GGGraph::ConfiguredParts();
	}
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &context, GGOpened &opened) {
		opened.graph = GGBuildGraph(context, spec);
		// (the planner's count(*) needs the number of walks only: degrees, not the counting expansion's checksum)
		opened.source = make_unique<PhysicalGGPathExpand>(opened.graph, hops, hops, count_only, sources, all_sources, 0,
		                                                  count_only);
	};
	data->description = pattern.edge_table->name + ": " + pattern.edge_table->columns[pattern.src_column].name +
	                    " -> " + pattern.edge_table->columns[pattern.dst_column].name + "\n" + to_string(hops) +
	                    (hops == 1 ? " hop" : " hops") + "\nvertices: " +
	                    (pattern.vertex_table ? pattern.vertex_table->name + "." +
	                                                pattern.vertex_table->columns[pattern.vertex_key].name
	                                          : string("endpoint ids")) +
	                    (all_sources ? string() : "\nfrom " + to_string(sources[0])) +
	                    (spec.shards > 1 ? "\nshards: " + to_string(spec.shards) : string());
	data->parallel_result = !count_only;
	auto types = PhysicalGGPathExpand::OutputTypes(hops, count_only);
	if (g_plan_context && GGPipelineSinksAvailable(*g_plan_context, spec)) {
		// the tables reach the device through pipeline sinks the reference's executor schedules (gg_pipeline.cpp)
		g_rules_fired++;
		return GGMakeGraphScan(
		    spec, move(types), count_only ? "GG_PATH_COUNT" : "GG_PATH_EXPAND", data->description, !count_only,
		    [=](ClientContext &, shared_ptr<GGGraph> graph) -> unique_ptr<PhysicalOperator> {
			    return make_unique<PhysicalGGPathExpand>(move(graph), hops, hops, count_only, sources, all_sources, 0,
			                                             count_only);
		    },
		    estimated_cardinality);
	}
	vector<column_t> column_ids;
	vector<string> names;
	for (idx_t c = 0; c < types.size(); c++) {
		column_ids.push_back(c);
		names.push_back("c" + to_string(c));
	}
	g_rules_fired++;
	return make_unique<PhysicalTableScan>(move(types), GGScanFunction(count_only ? "gg_path_count" : "gg_path_expand"),
	                                      move(data), move(column_ids), move(names), nullptr, estimated_cardinality);
}


//===--------------------------------------------------------------------===//
// Join rule 2: fixed-length walks whose vertices all share one neighbour in a second edge table
//===--------------------------------------------------------------------===//
// benchmark/trainbenchmark/queries/connectedsegments.sql:1-25 (BASELINE.json configs[4]):
//
//   FROM Segment JOIN connectsTo ct1 ON Segment.id = ct1.TE1 JOIN connectsTo ct2 ON ct1.TE2 = ct2.TE1 ... ct5
//        JOIN monitoredBy mb1 ON mb1.TE = ct1.TE1 ... JOIN monitoredBy mb6 ON mb6.TE = ct5.TE2
//   WHERE mb1.Sensor = mb2.Sensor AND ... AND mb1.Sensor = mb6.Sensor
//
// i.e. h path-edge instances forming a walk, h+1 filter-edge instances — one per walk position, keyed by
// it — whose far ends are all equal, and optionally one instance of a uniquely keyed table pinning where
// walks may start.  Eleven hash joins in the reference; here PhysicalGGFilteredPaths (k-hop expansion +
// gg_result_filter_common_neighbour).  Every column of the pattern is a join key, so NULLs drop out on
// both sides and no NOT NULL declaration is needed; the vertex set is the union of both tables' endpoint
// ids, so no row is dropped for a missing vertex either.
struct SameNeighbourPattern {
	TableCatalogEntry *path_table = nullptr, *filter_table = nullptr, *source_table = nullptr;
	column_t path_src = 0, path_dst = 0, filter_src = 0, filter_dst = 0, source_key = 0;
	idx_t hops = 0;
	// per leaf: what its columns mean
	vector<idx_t> path_position;   // 1-based edge number (0: not a path leaf)
	vector<idx_t> filter_position; // walk position the filter leaf is keyed by (INVALID_INDEX: not one)
	idx_t source_leaf = INVALID_INDEX;
};

bool SolveSameNeighbourWithRoles(PatternInput &in, ColumnClasses &classes, SameNeighbourPattern &out) {
	const idx_t n = in.leaves.size();
	vector<idx_t> path_leaves, filter_leaves;
	out.source_leaf = INVALID_INDEX;
	for (idx_t l = 0; l < n; l++) {
		if (in.leaves[l].table == out.path_table) {
			path_leaves.push_back(l);
		} else if (in.leaves[l].table == out.filter_table) {
			filter_leaves.push_back(l);
		} else if (in.leaves[l].table == out.source_table && out.source_leaf == INVALID_INDEX) {
			out.source_leaf = l;
		} else {
			return false;
		}
	}
	const idx_t hops = path_leaves.size();
	if (hops < 1 || hops + 1 > GG_MAX_HOPS || filter_leaves.size() != hops + 1) {
		return false;
	}
	// the walk: every position must be in an equality class (a filter leaf is keyed by it)
	vector<idx_t> src_class(n, INVALID_INDEX), dst_class(n, INVALID_INDEX), successor(n, INVALID_INDEX);
	vector<bool> has_predecessor(n, false);
	for (auto l : path_leaves) {
		src_class[l] = classes.ClassOf({l, out.path_src});
		dst_class[l] = classes.ClassOf({l, out.path_dst});
		if (src_class[l] == INVALID_INDEX || dst_class[l] == INVALID_INDEX || src_class[l] == dst_class[l]) {
			return false;
		}
	}
	for (auto a : path_leaves) {
		for (auto b : path_leaves) {
			if (a == b) {
				continue;
			}
			if (src_class[a] == src_class[b] || dst_class[a] == dst_class[b]) {
				return false;
			}
			if (dst_class[a] == src_class[b]) {
				successor[a] = b;
				has_predecessor[b] = true;
			}
		}
	}
	idx_t first = INVALID_INDEX;
	for (auto l : path_leaves) {
		if (!has_predecessor[l]) {
			if (first != INVALID_INDEX) {
				return false;
			}
			first = l;
		}
	}
	if (first == INVALID_INDEX) {
		return false;
	}
	out.path_position.assign(n, 0);
	out.filter_position.assign(n, INVALID_INDEX);
	vector<idx_t> position_class(hops + 1, INVALID_INDEX);
	idx_t visited = 0;
	for (idx_t l = first; l != INVALID_INDEX; l = successor[l]) {
		if (out.path_position[l] != 0 || visited == hops) {
			return false;
		}
		out.path_position[l] = ++visited;
		position_class[visited - 1] = src_class[l];
		position_class[visited] = dst_class[l];
	}
	if (visited != hops) {
		return false;
	}
	for (idx_t p = 0; p <= hops; p++) { // a walk position may not coincide with another one (no cycles pinned)
		for (idx_t q = p + 1; q <= hops; q++) {
			if (position_class[p] == position_class[q]) {
				return false;
			}
		}
	}
	// the filter leaves: keyed by distinct positions, far ends all in one class of their own
	idx_t neighbour_class = INVALID_INDEX;
	vector<bool> covered(hops + 1, false);
	for (auto l : filter_leaves) {
		const auto key = classes.ClassOf({l, out.filter_src});
		const auto far = classes.ClassOf({l, out.filter_dst});
		if (key == INVALID_INDEX || far == INVALID_INDEX) {
			return false;
		}
		if (neighbour_class == INVALID_INDEX) {
			neighbour_class = far;
		} else if (neighbour_class != far) {
			return false;
		}
		idx_t position = INVALID_INDEX;
		for (idx_t p = 0; p <= hops; p++) {
			if (position_class[p] == key) {
				position = p;
			}
		}
		if (position == INVALID_INDEX || covered[position]) {
			return false;
		}
		covered[position] = true;
		out.filter_position[l] = position;
	}
	for (idx_t p = 0; p <= hops; p++) {
		if (position_class[p] == neighbour_class) {
			return false;
		}
	}
	// the source table, if any: its key pins walk position 0
	if (out.source_leaf != INVALID_INDEX) {
		if (classes.ClassOf({out.source_leaf, out.source_key}) != position_class[0] ||
		    !ColumnIsUnique(*out.source_table, out.source_key) || !ColumnIsIntegerKey(*out.source_table, out.source_key)) {
			return false;
		}
	}
	// nothing else is equated: every member of every class has one of the roles above
	for (idx_t i = 0; i < classes.members.size(); i++) {
		auto &member = classes.members[i];
		const auto cls = classes.Find(i);
		bool ok;
		if (out.path_position[member.leaf]) {
			const idx_t e = out.path_position[member.leaf];
			ok = (member.column == out.path_src && cls == position_class[e - 1]) ||
			     (member.column == out.path_dst && cls == position_class[e]);
		} else if (out.filter_position[member.leaf] != INVALID_INDEX) {
			ok = (member.column == out.filter_src && cls == position_class[out.filter_position[member.leaf]]) ||
			     (member.column == out.filter_dst && cls == neighbour_class);
		} else {
			ok = member.leaf == out.source_leaf && member.column == out.source_key && cls == position_class[0];
		}
		if (!ok) {
			return false;
		}
	}
	for (auto column : {out.path_src, out.path_dst}) {
		if (!ColumnIsIntegerKey(*out.path_table, column)) {
			return false;
		}
	}
	for (auto column : {out.filter_src, out.filter_dst}) {
		if (!ColumnIsIntegerKey(*out.filter_table, column)) {
			return false;
		}
	}
	out.hops = hops;
	return true;
}

bool SolveSameNeighbourPattern(PatternInput &in, SameNeighbourPattern &out) {
	if (in.leaves.size() < 3 || !in.constants.empty() || !in.filters.empty() || !in.other_filters.empty()) {
		return false;
	}
	vector<TableCatalogEntry *> tables;
	for (auto &leaf : in.leaves) {
		if (std::find(tables.begin(), tables.end(), leaf.table) == tables.end()) {
			tables.push_back(leaf.table);
		}
	}
	if (tables.size() < 2 || tables.size() > 3) {
		return false;
	}
	ColumnClasses classes;
	for (auto &eq : in.equalities) {
		const auto a = classes.Add(eq.first), b = classes.Add(eq.second);
		classes.Union(a, b);
	}
	auto columns_used = [&](TableCatalogEntry *table) {
		vector<column_t> used;
		for (auto &member : classes.members) {
			if (in.leaves[member.leaf].table == table && std::find(used.begin(), used.end(), member.column) == used.end()) {
				used.push_back(member.column);
			}
		}
		std::sort(used.begin(), used.end());
		return used;
	};
	for (auto path_table : tables) {
		for (auto filter_table : tables) {
			if (filter_table == path_table) {
				continue;
			}
			TableCatalogEntry *source_table = nullptr;
			for (auto table : tables) {
				if (table != path_table && table != filter_table) {
					source_table = table;
				}
			}
			auto path_columns = columns_used(path_table), filter_columns = columns_used(filter_table);
			vector<column_t> source_columns;
			if (source_table) {
				source_columns = columns_used(source_table);
				if (source_columns.size() != 1) {
					continue;
				}
			}
			if (path_columns.size() != 2 || filter_columns.size() != 2) {
				continue;
			}
			for (int path_flip = 0; path_flip < 2; path_flip++) {
				for (int filter_flip = 0; filter_flip < 2; filter_flip++) {
					out.path_table = path_table;
					out.filter_table = filter_table;
					out.source_table = source_table;
					out.path_src = path_columns[path_flip];
					out.path_dst = path_columns[1 - path_flip];
					out.filter_src = filter_columns[filter_flip];
					out.filter_dst = filter_columns[1 - filter_flip];
					out.source_key = source_table ? source_columns[0] : 0;
					if (SolveSameNeighbourWithRoles(in, classes, out)) {
						return true;
					}
				}
			}
		}
	}
	return false;
}

unique_ptr<PhysicalOperator> PlanSameNeighbourPaths(LogicalComparisonJoin &op, PatternInput &in) {
	SameNeighbourPattern pattern;
	if (!SolveSameNeighbourPattern(in, pattern)) {
		return nullptr;
	}
	// scan columns: (w, v0, ..., vh), all BIGINT
	auto bindings = op.GetColumnBindings();
	if (bindings.size() != op.types.size()) {
		return nullptr;
	}
	vector<unique_ptr<Expression>> select_list;
	for (idx_t i = 0; i < bindings.size(); i++) {
		LeafColumn column;
		if (!ResolveLeafColumn(in, bindings[i], column)) {
			return nullptr;
		}
		idx_t scan_column;
		if (pattern.path_position[column.leaf]) {
			if (column.column == pattern.path_src) {
				scan_column = 1 + pattern.path_position[column.leaf] - 1;
			} else if (column.column == pattern.path_dst) {
				scan_column = 1 + pattern.path_position[column.leaf];
			} else {
				return nullptr;
			}
		} else if (pattern.filter_position[column.leaf] != INVALID_INDEX) {
			if (column.column == pattern.filter_src) {
				scan_column = 1 + pattern.filter_position[column.leaf];
			} else if (column.column == pattern.filter_dst) {
				scan_column = 0;
			} else {
				return nullptr;
			}
		} else if (column.leaf == pattern.source_leaf && column.column == pattern.source_key) {
			scan_column = 1;
		} else {
			return nullptr;
		}
		if (in.leaves[column.leaf].table->columns[column.column].type != op.types[i]) {
			return nullptr;
		}
		unique_ptr<Expression> ref = make_unique<BoundReferenceExpression>(LogicalType::BIGINT, scan_column);
		if (op.types[i] != LogicalType::BIGINT) {
			ref = make_unique<BoundCastExpression>(move(ref), op.types[i]);
		}
		select_list.push_back(move(ref));
	}
	const int hops = (int)pattern.hops;
	const auto path = TableColumns(pattern.path_table, {pattern.path_src, pattern.path_dst});
	const auto filter = TableColumns(pattern.filter_table, {pattern.filter_src, pattern.filter_dst});
	GGScanSource sources;
	if (pattern.source_table) {
		sources = TableColumns(pattern.source_table, {pattern.source_key});
	}
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &context, GGOpened &opened) {
		opened.graph = make_shared<GGGraph>(0);
		const vector<LogicalType> two = {LogicalType::BIGINT, LogicalType::BIGINT};
		// vertex set = endpoint ids of both edge tables: the path table contributes its endpoints first ...
		PhysicalGGEdgeSink endpoints(opened.graph, two, 0, false, true, false, false);
		GGRunSinkPipeline(context, path, endpoints);
		// ... the filter table adds its own and builds the filter CSR over the union ...
		PhysicalGGEdgeSink filter_sink(opened.graph, two, 0, true, true, true, true);
		GGRunSinkPipeline(context, filter, filter_sink);
		// ... and the path table comes back for its CSR over the same vertex numbering
		GGGraph::Check(gg_staging_clear_edges(opened.graph->ctx), "gg_staging_clear_edges");
		PhysicalGGEdgeSink path_sink(opened.graph, two, 0);
		GGRunSinkPipeline(context, path, path_sink);
		vector<int64_t> source_ids;
		if (sources.table) {
			source_ids = GGScanInt64Column(context, sources);
		}
		opened.source = make_unique<PhysicalGGFilteredPaths>(opened.graph, hops, move(source_ids), 0, sources.table == nullptr);
	};
	data->description = pattern.path_table->name + ": " + pattern.path_table->columns[pattern.path_src].name + " -> " +
	                    pattern.path_table->columns[pattern.path_dst].name + "\n" + to_string(hops) +
	                    (hops == 1 ? " hop" : " hops") + "\nall on one " + pattern.filter_table->name + "." +
	                    pattern.filter_table->columns[pattern.filter_dst].name +
	                    (pattern.source_table ? "\nfrom every " + pattern.source_table->name : string());
	vector<LogicalType> types(hops + 2, LogicalType::BIGINT);
	g_rules_fired++;
	unique_ptr<PhysicalOperator> scan;
	GGGraphSpec sinks_spec;  // (what GGPipelineSinksAvailable looks at: plain tables, no pinned graphs)
	sinks_spec.edges = path;
	if (g_plan_context && filter.table && GGPipelineSinksAvailable(*g_plan_context, sinks_spec)) {
		// the three table passes as pipeline sinks the reference's executor schedules (gg_pipeline.cpp), in build order:
		// endpoints of the path table, the filter table (its endpoints and its CSR), the path table's CSR
		vector<GGSinkSpec> sinks(3);
		sinks[0].rows = path;
		sinks[0].options.first = true;
		sinks[0].options.derive_vertices = true;
		sinks[0].options.build = false;
		sinks[0].options.clear_edges_after = true;  // only its endpoints were wanted: the filter CSR is built from the filter table's rows alone
		sinks[1].rows = filter;
		sinks[1].options.as_filter = sinks[1].options.derive_vertices = sinks[1].options.keep_vertices = true;
		sinks[1].options.clear_edges_after = true;  // the path table's rows come back for its own CSR
		sinks[2].rows = path;
		scan = GGMakeGraphScan(
		    sinks, move(types), "GG_SAME_NEIGHBOUR_WALKS", data->description, false /* one thread drains the result */,
		    [=](ClientContext &context, shared_ptr<GGGraph> graph) -> unique_ptr<PhysicalOperator> {
			    vector<int64_t> source_ids;
			    if (sources.table) {
				    source_ids = GGScanInt64Column(context, sources);
			    }
			    return make_unique<PhysicalGGFilteredPaths>(move(graph), hops, move(source_ids), 0, sources.table == nullptr);
		    },
		    op.estimated_cardinality);
	} else {
		vector<column_t> column_ids;
		vector<string> names;
		for (idx_t c = 0; c < types.size(); c++) {
			column_ids.push_back(c);
			names.push_back(c == 0 ? "w" : "v" + to_string(c - 1));
		}
		scan = make_unique<PhysicalTableScan>(move(types), GGScanFunction("gg_same_neighbour_walks"), move(data),
		                                      move(column_ids), move(names), nullptr, op.estimated_cardinality);
	}
	auto projection = make_unique<PhysicalProjection>(op.types, move(select_list), op.estimated_cardinality);
	projection->children.push_back(move(scan));
	return move(projection);
}

//! Join rule: the subtree's columns are walk vertices.
//! The same walks with payload columns of the edge table: the graph is built with the edges' rowids, the expansion
//! returns the rowid of every edge of every walk, PhysicalGGPathEdges fetches the columns by rowid (scan-function
//! route: the tables are read when the scan opens).
unique_ptr<PhysicalOperator> MakeEdgeScan(const WalkPattern &pattern, const vector<std::pair<idx_t, column_t>> &payload,
                                          idx_t estimated_cardinality) {
	auto spec = GraphSpecOf(pattern);
	spec.edges.columns.push_back(COLUMN_IDENTIFIER_ROW_ID);
	const int hops = (int)pattern.hops;
	const auto sources = pattern.sources;
	const bool all_sources = pattern.all_sources;
	auto edge_table = pattern.edge_table;
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &context, GGOpened &opened) {
		opened.graph = GGBuildGraph(context, spec);
		opened.source = make_unique<PhysicalGGPathEdges>(opened.graph, hops, sources, all_sources, edge_table, payload, 0);
	};
	data->description = pattern.edge_table->name + ": " + pattern.edge_table->columns[pattern.src_column].name + " -> " +
	                    pattern.edge_table->columns[pattern.dst_column].name + "\n" + to_string(hops) +
	                    (hops == 1 ? " hop" : " hops") + "\nwith " + to_string(payload.size()) +
	                    (payload.size() == 1 ? " edge column by rowid" : " edge columns by rowid") +
	                    (all_sources ? string() : "\nfrom " + to_string(sources[0]));
	auto types = PhysicalGGPathEdges::OutputTypes(hops, *edge_table, payload);
	vector<column_t> column_ids;
	vector<string> names;
	for (idx_t c = 0; c < types.size(); c++) {
		column_ids.push_back(c);
		names.push_back("c" + to_string(c));
	}
	g_rules_fired++;
	return make_unique<PhysicalTableScan>(move(types), GGScanFunction("gg_path_edges"), move(data), move(column_ids),
	                                      move(names), nullptr, estimated_cardinality);
}

unique_ptr<PhysicalOperator> PlanJoinChain(LogicalComparisonJoin &op) {
	PatternInput in;
	WalkPattern pattern;
	if (!CollectJoinTree(op, in)) {
		return nullptr;
	}
	if (!SolveWalkPattern(in, pattern)) {
		return PlanSameNeighbourPaths(op, in);
	}
	// output layout of the join: map every column to a walk position
	auto bindings = op.GetColumnBindings();
	if (bindings.size() != op.types.size()) {
		return nullptr;
	}
	vector<unique_ptr<Expression>> select_list;
	//! columns of edge instances other than the two keys: (1-based edge number, column) — fetched by rowid behind the
	//! expansion (PhysicalGGPathEdges); scan column 2 + hops + index
	vector<std::pair<idx_t, column_t>> payload;
	for (idx_t i = 0; i < bindings.size(); i++) {
		LeafColumn column;
		if (!ResolveLeafColumn(in, bindings[i], column)) {
			return nullptr;
		}
		idx_t position;
		auto &table = *in.leaves[column.leaf].table;
		if (pattern.edge_position[column.leaf]) {
			if (column.column == pattern.src_column) {
				position = pattern.edge_position[column.leaf] - 1;
			} else if (column.column == pattern.dst_column) {
				position = pattern.edge_position[column.leaf];
			} else {
				if (table.columns[column.column].type != op.types[i] || pattern.hops > 4) {
					return nullptr;
				}
				const auto entry = std::make_pair(pattern.edge_position[column.leaf], (column_t)column.column);
				idx_t at = std::find(payload.begin(), payload.end(), entry) - payload.begin();
				if (at == payload.size()) {
					payload.push_back(entry);
				}
				select_list.push_back(make_unique<BoundReferenceExpression>(op.types[i], 2 + pattern.hops + at));
				continue;
			}
		} else {
			if (column.column != pattern.vertex_key) {
				return nullptr;
			}
			position = pattern.vertex_position[column.leaf];
		}
		if (table.columns[column.column].type != op.types[i]) {
			return nullptr;
		}
		// scan columns: (hops, v0, v1, ...) all BIGINT; narrower key columns get their type back
		unique_ptr<Expression> ref = make_unique<BoundReferenceExpression>(LogicalType::BIGINT, 1 + position);
		if (op.types[i] != LogicalType::BIGINT) {
			ref = make_unique<BoundCastExpression>(move(ref), op.types[i]);
		}
		select_list.push_back(move(ref));
	}
	// predicates on payload columns of the edge instances: the columns join the fetched ones (whether or not the
	// statement projects them) and the predicate is evaluated above the scan, on every walk with ITS edges' values
	vector<unique_ptr<Expression>> payload_predicates;
	for (auto &entry : pattern.payload_filters) {
		if (pattern.hops > 4) {
			return nullptr;
		}
		const auto key = std::make_pair(entry.edge, entry.column);
		idx_t at = std::find(payload.begin(), payload.end(), key) - payload.begin();
		if (at == payload.size()) {
			payload.push_back(key);
		}
		payload_predicates.push_back(PayloadFilterToExpression(*entry.filter, 2 + pattern.hops + at,
		                                                       pattern.edge_table->columns[entry.column].type));
	}
	if (!payload.empty() && (!g_plan_context || Transaction::GetTransaction(*g_plan_context).ChangesMade())) {
		return nullptr; // (rows this transaction has appended have no rowid the base table could be asked for)
	}
	auto scan = payload.empty() ? MakeExpandScan(pattern, false, op.estimated_cardinality)
	                            : MakeEdgeScan(pattern, payload, op.estimated_cardinality);
	if (!pattern.residual.empty() || !payload_predicates.empty()) {
		// predicates on walk positions other than the source: a filter over the scan's (hops, v0, v1, ...)
		vector<unique_ptr<Expression>> predicates = move(payload_predicates);
		for (auto &entry : pattern.residual) {
			predicates.push_back(FilterToExpression(*entry.second, 1 + entry.first));
		}
		auto filter = make_unique<PhysicalFilter>(scan->types, move(predicates), op.estimated_cardinality);
		filter->children.push_back(move(scan));
		scan = move(filter);
	}
	auto projection = make_unique<PhysicalProjection>(op.types, move(select_list), op.estimated_cardinality);
	projection->children.push_back(move(scan));
	return move(projection);
}

//! count(*) over ONE inner equi-join of two base tables on an integer key, `probe.x = build.y` with duplicates on
//! either side (test/sql/join/inner/test_join_duplicates.test:14-24: 10 240 build rows under one key): the build table
//! is the adjacency index keyed on y (every row an entry under its key), the probe table's x column is the source
//! list, and the join's cardinality is the number of 1-hop walks from it — the sum over the probe rows of the degree
//! of their key (gg_khop_count), as PhysicalHashJoin's probe would emit and count them
//! (src/execution/join_hashtable.cpp:304-476).  NULL keys join nothing on either side (the sinks skip them).
unique_ptr<PhysicalOperator> MakeKeyJoinCountScan(PatternInput &in) {
	if (in.leaves.size() != 2 || in.equalities.size() != 1 || !in.constants.empty() || !in.filters.empty() ||
	    !in.other_filters.empty() || !in.aliases.empty()) {
		return nullptr;
	}
	auto a = in.equalities[0].first, b = in.equalities[0].second;
	if (a.leaf == b.leaf) {
		return nullptr;
	}
	if (a.leaf != 0) {
		std::swap(a, b);
	}
	auto probe_table = in.leaves[a.leaf].table, build_table = in.leaves[b.leaf].table;
	if (!ColumnIsIntegerKey(*probe_table, a.column) || !ColumnIsIntegerKey(*build_table, b.column)) {
		return nullptr;
	}
	GGGraphSpec spec;
	spec.edges = TableColumns(build_table, {b.column, b.column}); // (key -> key: only the degree of a key matters)
	const auto probe = TableColumns(probe_table, {a.column});
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &context, GGOpened &opened) {
		opened.graph = GGBuildGraph(context, spec);
		opened.source = make_unique<PhysicalGGPathExpand>(opened.graph, 1, 1, true, GGScanInt64Column(context, probe), false,
		                                                  0, true);
	};
	data->description = build_table->name + "." + build_table->columns[b.column].name + "\nprobed with " +
	                    probe_table->name + "." + probe_table->columns[a.column].name + "\n1 hop";
	data->parallel_result = false;
	auto types = PhysicalGGPathExpand::OutputTypes(1, true);
	g_rules_fired++;
	if (g_plan_context && GGPipelineSinksAvailable(*g_plan_context, spec)) {
		return GGMakeGraphScan(
		    spec, move(types), "GG_JOIN_COUNT", data->description, false,
		    [=](ClientContext &context, shared_ptr<GGGraph> graph) -> unique_ptr<PhysicalOperator> {
			    return make_unique<PhysicalGGPathExpand>(move(graph), 1, 1, true, GGScanInt64Column(context, probe), false, 0,
			                                             true);
		    },
		    1);
	}
	vector<column_t> column_ids;
	vector<string> names;
	for (idx_t c = 0; c < types.size(); c++) {
		column_ids.push_back(c);
		names.push_back("c" + to_string(c));
	}
	return make_unique<PhysicalTableScan>(move(types), GGScanFunction("gg_join_count"), move(data), move(column_ids),
	                                      move(names), nullptr, 1);
}

//! Join rule 3 (PRAGMA enable_gpu_joins): ANY inner join on one equality of two integer columns whose build side —
//! the right child, as the reference's own planner has it (plan_comparison_join.cpp:146-220) — is a plain scan of a
//! base table.  The build side is sunk into a device index keyed on its column, the probe side (any plan: the
//! generator plans it) streams through PhysicalGGKeyJoin.  This is the reference's PhysicalHashJoin role for role, at
//! one device round trip per probe chunk — what makes the operators a drop-in for joins that are not walks (the
//! reference's own join vectors, a hash join in the arm of a recursive CTE), not a fast path: off unless asked for.
unique_ptr<PhysicalOperator> PlanKeyJoin(LogicalComparisonJoin &op) {
	if (!g_plan_context || !g_plan_generator || !GGGetConnectionFlags(*g_plan_context).joins) {
		if (std::getenv("GG_RULE_TRACE")) fprintf(stderr, "[gg] key join declined at check 1\n");
		return nullptr;
	}
	if (op.join_type != JoinType::INNER || op.conditions.size() != 1 || op.children.size() != 2) {
		if (std::getenv("GG_RULE_TRACE")) fprintf(stderr, "[gg] key join declined at check 2\n");
		return nullptr;
	}
	auto &cond = op.conditions[0];
	if (cond.comparison != ExpressionType::COMPARE_EQUAL || cond.null_values_are_equal ||
	    cond.left->type != ExpressionType::BOUND_REF || cond.right->type != ExpressionType::BOUND_REF) {
		if (std::getenv("GG_RULE_TRACE")) fprintf(stderr, "[gg] key join declined at check 3\n");
		return nullptr;
	}
	if (op.children[1]->type != LogicalOperatorType::LOGICAL_GET) {
		if (std::getenv("GG_RULE_TRACE")) fprintf(stderr, "[gg] key join declined at check 4\n");
		return nullptr;
	}
	auto &get = (LogicalGet &)*op.children[1];
	if (!get.children.empty() || get.function.name != "seq_scan" || !get.bind_data || !get.table_filters.filters.empty()) {
		if (std::getenv("GG_RULE_TRACE")) fprintf(stderr, "[gg] key join declined at check 5\n");
		return nullptr;
	}
	auto &bind = (TableScanBindData &)*get.bind_data;
	if (bind.is_index_scan || !bind.table) {
		if (std::getenv("GG_RULE_TRACE")) fprintf(stderr, "[gg] key join declined at check 6\n");
		return nullptr;
	}
	auto table = bind.table;
	const idx_t probe_key = ((BoundReferenceExpression &)*cond.left).index;
	const idx_t build_ref = ((BoundReferenceExpression &)*cond.right).index;
	if (build_ref >= get.column_ids.size() || get.column_ids[build_ref] == COLUMN_IDENTIFIER_ROW_ID ||
	    probe_key >= op.children[0]->types.size()) {
		if (std::getenv("GG_RULE_TRACE")) fprintf(stderr, "[gg] key join declined at check 7\n");
		return nullptr;
	}
	const column_t build_key = get.column_ids[build_ref];
	const auto key_type = cond.left->return_type;
	if (!ColumnIsIntegerKey(*table, build_key) || key_type != table->columns[build_key].type ||
	    (key_type != LogicalType::BIGINT && key_type != LogicalType::INTEGER)) {
		if (std::getenv("GG_RULE_TRACE")) fprintf(stderr, "[gg] key join declined at check 8\n");
		return nullptr;
	}
	// rows this transaction appended itself have no fetchable rowid yet (DataTable::Fetch): its own hash join then
	if (Transaction::GetTransaction(*g_plan_context).ChangesMade()) {
		if (std::getenv("GG_RULE_TRACE")) fprintf(stderr, "[gg] key join declined at check 9\n");
		return nullptr;
	}
	const auto types = op.types;
	const auto cardinality = op.estimated_cardinality;
	// the join's output: the probe child's columns, then the scan's — each through its projection map if column
	// pruning left one (LogicalComparisonJoin::ResolveTypes: an empty map keeps every column)
	vector<idx_t> probe_columns = op.left_projection_map;
	if (probe_columns.empty()) {
		for (idx_t c = 0; c < op.children[0]->types.size(); c++) {
			probe_columns.push_back(c);
		}
	}
	vector<column_t> build_columns;
	if (op.right_projection_map.empty()) {
		build_columns = get.column_ids;
	} else {
		for (auto c : op.right_projection_map) {
			if (c >= get.column_ids.size()) {
				return nullptr;
			}
			build_columns.push_back(get.column_ids[c]);
		}
	}
	if (probe_columns.size() + build_columns.size() != op.types.size()) {
		return nullptr;
	}
	auto build_scan = GGBaseTableScan(TableColumns(table, {build_key, build_key, COLUMN_IDENTIFIER_ROW_ID}));
	// (from here on the logical join is spent: its probe child moves into the generator — which may come back into
	//  these rules for joins further down, with the per-plan bookkeeping of its own)
	auto generator = g_plan_generator;
	auto probe = generator->CreatePlan(move(op.children[0]));
	g_plan_generator = generator;
	g_plan_tables.push_back(table);
	g_rules_fired++;
	return make_unique<PhysicalGGKeyJoin>(types, move(probe), move(build_scan), probe_key, move(probe_columns), table,
	                                      build_key, move(build_columns), cardinality);
}

unique_ptr<PhysicalOperator> PlanJoin(LogicalComparisonJoin &op) {
	if (auto plan = PlanJoinChain(op)) {
		return plan;
	}
	return PlanKeyJoin(op);
}

//! Aggregate rule: ungrouped count(*) over a walk pattern.
unique_ptr<PhysicalOperator> PlanCountOverJoinChain(LogicalAggregate &op) {
	if (!op.groups.empty() || !op.grouping_functions.empty() || op.grouping_sets.size() > 1 || op.expressions.empty() ||
	    op.children.size() != 1) {
		return nullptr;
	}
	for (auto &expr : op.expressions) {
		if (expr->GetExpressionClass() != ExpressionClass::BOUND_AGGREGATE) {
			return nullptr;
		}
		auto &aggr = (BoundAggregateExpression &)*expr;
		if (aggr.function.name != "count_star" || aggr.distinct || aggr.filter || !aggr.children.empty()) {
			return nullptr;
		}
	}
	// projections between the aggregate and the join do not change the row count
	auto child = op.children[0].get();
	while (child->type == LogicalOperatorType::LOGICAL_PROJECTION && child->children.size() == 1) {
		child = child->children[0].get();
	}
	if (child->type != LogicalOperatorType::LOGICAL_COMPARISON_JOIN) {
		return nullptr;
	}
	PatternInput in;
	WalkPattern pattern;
	if (!CollectJoinTree(*child, in)) {
		return nullptr;
	}
	const bool is_walk = SolveWalkPattern(in, pattern);
	if (is_walk && (!pattern.residual.empty() || !pattern.payload_filters.empty())) {
		return nullptr; // (with a residual predicate the walks must be looked at: the join rule takes the join)
	}
	// scan columns: (hops, rows, digest, traversed_edges), one row; every count(*) is `rows`
	vector<unique_ptr<Expression>> select_list;
	for (idx_t i = 0; i < op.expressions.size(); i++) {
		if (op.types[i] != LogicalType::BIGINT) {
			return nullptr;
		}
		select_list.push_back(make_unique<BoundReferenceExpression>(LogicalType::BIGINT, 1));
	}
	if (op.types.size() != op.expressions.size()) {
		return nullptr;
	}
	auto scan = is_walk ? MakeExpandScan(pattern, true, 1) : MakeKeyJoinCountScan(in);
	if (!scan) {
		return nullptr;
	}
	auto projection = make_unique<PhysicalProjection>(op.types, move(select_list), 1);
	projection->children.push_back(move(scan));
	return move(projection);
}


//===--------------------------------------------------------------------===//
// Aggregate rule 2: min(hop) GROUP BY (start, friend) over the friends recursive CTE -> 64-lane BFS
//===--------------------------------------------------------------------===//
// benchmark/ldbc/queries/bi-10-shortestpath.sql:8-31 (friends / friends_shortest):
//
//   WITH RECURSIVE friends(startPerson, hopCount, friend) AS (
//       SELECT p_personid, 0, p_personid FROM person WHERE p_personid = <id>          -- seed
//     UNION
//       SELECT f.startPerson, f.hopCount+1, <k.k_person2id>                            -- step
//         FROM friends f, knows k [, person p]
//        WHERE f.friend = k.k_person1id [AND k.k_person2id = p.p_personid] AND f.hopCount < <K>)
//   SELECT startPerson, min(hopCount), friend FROM friends GROUP BY startPerson, friend
//
// The reference runs PhysicalRecursiveCTE (hash-join rebuild per level + UNION dedupe table) under a
// PhysicalHashAggregate; min over the levels at which a vertex shows up is its BFS distance, so the whole
// subtree is one PhysicalGGShortestPath.  The CTE alone is NOT replaced: it also holds the longer walks.
// Column order of the CTE, the seed predicate (none, `= c`, `IN (...)`, ORs of `= c`) and the optional
// validating join with the vertex table are read off the plan; anything else is declined.

//! a column somewhere below: (table index of the producing leaf, column id there)
struct PlanColumn {
	idx_t table_index = INVALID_INDEX;
	idx_t column = INVALID_INDEX;
	bool operator==(const PlanColumn &o) const {
		return table_index == o.table_index && column == o.column;
	}
};

struct StepInput {
	LogicalCTERef *cte = nullptr;
	int64_t max_hops = -1; // from `hop < K` on the CTE scan
	idx_t bound_column = INVALID_INDEX;
	vector<LogicalGet *> scans;
	vector<std::pair<PlanColumn, PlanColumn>> equalities;
};

bool IntegerConstant(Expression &expr, int64_t &value) {
	if (expr.type != ExpressionType::VALUE_CONSTANT) {
		return false;
	}
	auto &constant = ((BoundConstantExpression &)expr).value;
	if (constant.is_null || !constant.type().IsIntegral()) {
		return false;
	}
	value = constant.GetValue<int64_t>();
	return true;
}

bool ReferenceIndex(Expression &expr, idx_t &index) {
	if (expr.type != ExpressionType::BOUND_REF) {
		return false;
	}
	index = ((BoundReferenceExpression &)expr).index;
	return true;
}

//! binding of a leaf -> PlanColumn (GET: table column id; CTE scan: column position)
bool ResolvePlanColumn(StepInput &in, const ColumnBinding &binding, PlanColumn &out) {
	if (in.cte && in.cte->table_index == binding.table_index) {
		out.table_index = binding.table_index;
		out.column = binding.column_index;
		return true;
	}
	for (auto get : in.scans) {
		if (get->table_index == binding.table_index) {
			if (binding.column_index >= get->column_ids.size() ||
			    get->column_ids[binding.column_index] == COLUMN_IDENTIFIER_ROW_ID) {
				return false;
			}
			out.table_index = binding.table_index;
			out.column = get->column_ids[binding.column_index];
			return true;
		}
	}
	return false;
}

bool CollectStep(LogicalOperator &op, idx_t cte_index, StepInput &in) {
	switch (op.type) {
	case LogicalOperatorType::LOGICAL_CTE_REF: {
		auto &ref = (LogicalCTERef &)op;
		if (in.cte || ref.cte_index != cte_index) {
			return false;
		}
		in.cte = &ref;
		return true;
	}
	case LogicalOperatorType::LOGICAL_FILTER: {
		// only `hop < K` (or <=) directly on the CTE scan
		auto &filter = (LogicalFilter &)op;
		if (filter.children.size() != 1 || filter.children[0]->type != LogicalOperatorType::LOGICAL_CTE_REF ||
		    filter.expressions.size() != 1 || !filter.projection_map.empty() ||
		    !CollectStep(*filter.children[0], cte_index, in)) {
			return false;
		}
		auto &expr = *filter.expressions[0];
		if (expr.type != ExpressionType::COMPARE_LESSTHAN && expr.type != ExpressionType::COMPARE_LESSTHANOREQUALTO) {
			return false;
		}
		auto &cmp = (BoundComparisonExpression &)expr;
		int64_t bound;
		if (!ReferenceIndex(*cmp.left, in.bound_column) || !IntegerConstant(*cmp.right, bound)) {
			return false;
		}
		in.max_hops = expr.type == ExpressionType::COMPARE_LESSTHAN ? bound : bound + 1;
		return in.max_hops >= 0 && in.max_hops < (1 << 30);
	}
	case LogicalOperatorType::LOGICAL_GET: {
		auto &get = (LogicalGet &)op;
		if (!get.children.empty() || get.function.name != "seq_scan" || !get.table_filters.filters.empty() ||
		    !get.bind_data || ((TableScanBindData &)*get.bind_data).is_index_scan) {
			return false;
		}
		in.scans.push_back(&get);
		return true;
	}
	case LogicalOperatorType::LOGICAL_COMPARISON_JOIN: {
		auto &join = (LogicalComparisonJoin &)op;
		if (join.join_type != JoinType::INNER || join.children.size() != 2 ||
		    !CollectStep(*join.children[0], cte_index, in) || !CollectStep(*join.children[1], cte_index, in)) {
			return false;
		}
		auto left_bindings = join.children[0]->GetColumnBindings();
		auto right_bindings = join.children[1]->GetColumnBindings();
		for (auto &cond : join.conditions) {
			idx_t li, ri;
			PlanColumn l, r;
			if (cond.comparison != ExpressionType::COMPARE_EQUAL || cond.null_values_are_equal ||
			    !ReferenceIndex(*cond.left, li) || !ReferenceIndex(*cond.right, ri) || li >= left_bindings.size() ||
			    ri >= right_bindings.size() || !ResolvePlanColumn(in, left_bindings[li], l) ||
			    !ResolvePlanColumn(in, right_bindings[ri], r)) {
				return false;
			}
			in.equalities.emplace_back(l, r);
		}
		return true;
	}
	default:
		return false;
	}
}

//! The seed's predicate over the key: nothing, `key = c`, `key IN (c...)`, `key = c1 OR key = c2 ...`
bool SeedConstants(Expression &expr, idx_t key_index, vector<int64_t> &out) {
	idx_t index;
	int64_t value;
	switch (expr.type) {
	case ExpressionType::COMPARE_EQUAL: {
		auto &cmp = (BoundComparisonExpression &)expr;
		if (ReferenceIndex(*cmp.left, index) && index == key_index && IntegerConstant(*cmp.right, value)) {
			out.push_back(value);
			return true;
		}
		if (ReferenceIndex(*cmp.right, index) && index == key_index && IntegerConstant(*cmp.left, value)) {
			out.push_back(value);
			return true;
		}
		return false;
	}
	case ExpressionType::COMPARE_IN: {
		auto &in = (BoundOperatorExpression &)expr;
		if (in.children.size() < 2 || !ReferenceIndex(*in.children[0], index) || index != key_index) {
			return false;
		}
		for (idx_t i = 1; i < in.children.size(); i++) {
			if (!IntegerConstant(*in.children[i], value)) {
				return false;
			}
			out.push_back(value);
		}
		return true;
	}
	case ExpressionType::CONJUNCTION_OR: {
		for (auto &child : ((BoundConjunctionExpression &)expr).children) {
			if (!SeedConstants(*child, key_index, out)) {
				return false;
			}
		}
		return true;
	}
	default:
		return false;
	}
}

unique_ptr<PhysicalOperator> PlanShortestPath(LogicalAggregate &op) {
	// ---- the aggregate: GROUP BY two columns, min over a third
	idx_t group_ref[2], hop_ref;
	if (op.groups.size() != 2 || op.expressions.size() != 1 || !op.grouping_functions.empty() ||
	    op.grouping_sets.size() > 1 || op.children.size() != 1 || !ReferenceIndex(*op.groups[0], group_ref[0]) ||
	    !ReferenceIndex(*op.groups[1], group_ref[1]) ||
	    op.expressions[0]->GetExpressionClass() != ExpressionClass::BOUND_AGGREGATE) {
		return nullptr;
	}
	auto &aggr = (BoundAggregateExpression &)*op.expressions[0];
	if (aggr.function.name != "min" || aggr.distinct || aggr.filter || aggr.children.size() != 1 ||
	    !ReferenceIndex(*aggr.children[0], hop_ref)) {
		return nullptr;
	}
	// ---- pure column projections down to the recursive CTE
	auto node = op.children[0].get();
	while (node->type == LogicalOperatorType::LOGICAL_PROJECTION && node->children.size() == 1) {
		for (idx_t *ref : {&group_ref[0], &group_ref[1], &hop_ref}) {
			if (*ref >= node->expressions.size() || !ReferenceIndex(*node->expressions[*ref], *ref)) {
				return nullptr;
			}
		}
		node = node->children[0].get();
	}
	if (node->type != LogicalOperatorType::LOGICAL_RECURSIVE_CTE) {
		return nullptr;
	}
	auto &cte = (LogicalRecursiveCTE &)*node;
	if (cte.union_all || cte.column_count != 3 || group_ref[0] == group_ref[1] || hop_ref == group_ref[0] ||
	    hop_ref == group_ref[1] || group_ref[0] > 2 || group_ref[1] > 2 || hop_ref > 2) {
		return nullptr;
	}
	const idx_t hop_col = hop_ref;

	// ---- seed: SELECT key, 0, key FROM vertex_table [WHERE key ...]
	auto &seed = *cte.children[0];
	if (seed.type != LogicalOperatorType::LOGICAL_PROJECTION || seed.expressions.size() != 3 ||
	    seed.children.size() != 1) {
		return nullptr;
	}
	int64_t zero;
	idx_t key_ref[2];
	if (!IntegerConstant(*seed.expressions[hop_col], zero) || zero != 0 ||
	    seed.expressions[hop_col]->return_type != LogicalType::INTEGER ||
	    !ReferenceIndex(*seed.expressions[group_ref[0]], key_ref[0]) ||
	    !ReferenceIndex(*seed.expressions[group_ref[1]], key_ref[1]) || key_ref[0] != key_ref[1]) {
		return nullptr;
	}
	auto seed_child = seed.children[0].get();
	bool all_vertices = true;
	vector<int64_t> seed_constants;
	idx_t key_index = key_ref[0];
	// pure column projections (the IN-clause rewrite leaves one that drops its mark column)
	while (seed_child->type == LogicalOperatorType::LOGICAL_PROJECTION && seed_child->children.size() == 1) {
		if (key_index >= seed_child->expressions.size() ||
		    !ReferenceIndex(*seed_child->expressions[key_index], key_index)) {
			return nullptr;
		}
		seed_child = seed_child->children[0].get();
	}
	if (seed_child->type == LogicalOperatorType::LOGICAL_FILTER) {
		auto &filter = (LogicalFilter &)*seed_child;
		if (filter.expressions.size() != 1 || filter.children.size() != 1) {
			return nullptr;
		}
		if (!filter.projection_map.empty()) {
			if (key_index >= filter.projection_map.size()) {
				return nullptr;
			}
			key_index = filter.projection_map[key_index];
		}
		all_vertices = false;
		seed_child = filter.children[0].get();
		idx_t mark_index;
		if (seed_child->type == LogicalOperatorType::LOGICAL_COMPARISON_JOIN &&
		    ReferenceIndex(*filter.expressions[0], mark_index)) {
			// a long IN list: InClauseRewriter turned it into a MARK join against a constant chunk
			// (src/optimizer/in_clause_rewriter.cpp): FILTER(mark) <- MARK JOIN(scan, CHUNK_GET)
			auto &mark_join = (LogicalComparisonJoin &)*seed_child;
			idx_t left_ref, right_ref;
			if (mark_join.join_type != JoinType::MARK || mark_join.children.size() != 2 ||
			    mark_join.conditions.size() != 1 || !mark_join.left_projection_map.empty() ||
			    mark_join.conditions[0].comparison != ExpressionType::COMPARE_EQUAL ||
			    mark_join.children[1]->type != LogicalOperatorType::LOGICAL_CHUNK_GET ||
			    mark_index != mark_join.children[0]->GetColumnBindings().size() ||
			    !ReferenceIndex(*mark_join.conditions[0].left, left_ref) || left_ref != key_index ||
			    !ReferenceIndex(*mark_join.conditions[0].right, right_ref) || right_ref != 0) {
				return nullptr;
			}
			auto &chunk_get = (LogicalChunkGet &)*mark_join.children[1];
			if (!chunk_get.collection || chunk_get.collection->ColumnCount() != 1 ||
			    !chunk_get.chunk_types[0].IsIntegral()) {
				return nullptr;
			}
			for (idx_t r = 0; r < chunk_get.collection->Count(); r++) {
				auto value = chunk_get.collection->GetValue(0, r);
				if (!value.is_null) { // `key IN (..., NULL)` selects the same rows as without the NULL
					seed_constants.push_back(value.GetValue<int64_t>());
				}
			}
			seed_child = mark_join.children[0].get();
		} else if (!SeedConstants(*filter.expressions[0], key_index, seed_constants)) {
			return nullptr;
		}
	}
	if (seed_child->type != LogicalOperatorType::LOGICAL_GET) {
		return nullptr;
	}
	auto &seed_get = (LogicalGet &)*seed_child;
	if (!seed_get.children.empty() || seed_get.function.name != "seq_scan" || !seed_get.bind_data ||
	    key_index >= seed_get.column_ids.size()) {
		return nullptr;
	}
	// `key = c` on an indexed key turns the scan into an index scan but keeps the filter above it
	// (TableScanPushdownComplexFilter, src/function/table/table_scan.cpp): fine, the seed is re-read by key
	if (((TableScanBindData &)*seed_get.bind_data).is_index_scan && all_vertices) {
		return nullptr;
	}
	auto vertex_table = ((TableScanBindData &)*seed_get.bind_data).table;
	const auto vertex_key = seed_get.column_ids[key_index];
	if (vertex_key == COLUMN_IDENTIFIER_ROW_ID || !ColumnIsIntegerKey(*vertex_table, vertex_key)) {
		return nullptr;
	}
	// filters pushed into the seed's scan (`key = c`, or the bounds the optimizer derives from an IN list)
	vector<string> seed_predicates;
	vector<KeyPredicate> seed_filters;
	for (auto &entry : seed_get.table_filters.filters) {
		string predicate;
		KeyPredicate key_predicate;
		if (entry.first != vertex_key ||
		    !FilterToSQL(*entry.second, GGQuote(vertex_table->columns[vertex_key].name), predicate) ||
		    !FilterToPredicate(*entry.second, key_predicate)) {
			return nullptr;
		}
		seed_predicates.push_back(predicate);
		seed_filters.push_back(move(key_predicate));
	}

	// ---- step: SELECT f.start, f.hop + 1, <next vertex> FROM friends f JOIN edge [JOIN vertex] ...
	auto &step = *cte.children[1];
	if (step.type != LogicalOperatorType::LOGICAL_PROJECTION || step.expressions.size() != 3 ||
	    step.children.size() != 1) {
		return nullptr;
	}
	StepInput in;
	if (!CollectStep(*step.children[0], cte.table_index, in) || !in.cte || in.max_hops < 0 ||
	    in.bound_column != hop_col || in.scans.empty() || in.scans.size() > 2) {
		return nullptr;
	}
	auto step_bindings = step.children[0]->GetColumnBindings();
	auto column_of = [&](Expression &expr, PlanColumn &out) {
		idx_t index;
		return ReferenceIndex(expr, index) && index < step_bindings.size() &&
		       ResolvePlanColumn(in, step_bindings[index], out);
	};
	// hop + 1
	{
		auto &expr = *step.expressions[hop_col];
		if (expr.GetExpressionClass() != ExpressionClass::BOUND_FUNCTION) {
			return nullptr;
		}
		auto &plus = (BoundFunctionExpression &)expr;
		int64_t one;
		PlanColumn hop;
		if (plus.function.name != "+" || plus.children.size() != 2) {
			return nullptr;
		}
		const bool forward = column_of(*plus.children[0], hop) && IntegerConstant(*plus.children[1], one);
		const bool backward = !forward && column_of(*plus.children[1], hop) && IntegerConstant(*plus.children[0], one);
		if ((!forward && !backward) || one != 1 || hop.table_index != in.cte->table_index || hop.column != hop_col) {
			return nullptr;
		}
	}
	// which group column is passed through (start) and which advances (friend)
	idx_t start_col = INVALID_INDEX, friend_col = INVALID_INDEX;
	for (int g = 0; g < 2; g++) {
		PlanColumn passed;
		if (column_of(*step.expressions[group_ref[g]], passed) && passed.table_index == in.cte->table_index &&
		    passed.column == group_ref[g]) {
			start_col = group_ref[g];
			friend_col = group_ref[1 - g];
		}
	}
	if (start_col == INVALID_INDEX) {
		return nullptr;
	}
	// leaves: one edge scan, optionally one scan of the vertex table validating the new vertex
	LogicalGet *edge_get = nullptr, *validate_get = nullptr;
	for (auto get : in.scans) {
		auto table = ((TableScanBindData &)*get->bind_data).table;
		if (table == vertex_table && in.scans.size() == 2 && !validate_get) {
			validate_get = get;
		} else if (!edge_get) {
			edge_get = get;
		} else {
			return nullptr;
		}
	}
	if (!edge_get || (in.scans.size() == 2 && !validate_get)) {
		return nullptr;
	}
	auto edge_table = ((TableScanBindData &)*edge_get->bind_data).table;
	// equalities: f.friend = e.src  [and e.dst = p.key]
	PlanColumn cte_friend;
	cte_friend.table_index = in.cte->table_index;
	cte_friend.column = friend_col;
	column_t src = INVALID_INDEX, dst = INVALID_INDEX;
	bool validated = false;
	for (auto &eq : in.equalities) {
		for (int flip = 0; flip < 2; flip++) {
			auto &a = flip ? eq.second : eq.first;
			auto &b = flip ? eq.first : eq.second;
			if (a == cte_friend && b.table_index == edge_get->table_index && src == INVALID_INDEX) {
				src = b.column;
				goto next_equality;
			}
			if (validate_get && a.table_index == edge_get->table_index && b.table_index == validate_get->table_index &&
			    b.column == vertex_key && !validated) {
				dst = a.column;
				validated = true;
				goto next_equality;
			}
		}
		return nullptr; // an equality that is not part of the pattern
	next_equality:;
	}
	if (src == INVALID_INDEX || (validate_get != nullptr) != validated) {
		return nullptr;
	}
	// the advancing column: e.dst, the validated p.key, or bi-10's CASE WHEN f.friend = e.src THEN e.dst ELSE e.src
	{
		auto &expr = *step.expressions[friend_col];
		PlanColumn next;
		if (column_of(expr, next)) {
			if (validate_get && next.table_index == validate_get->table_index && next.column == vertex_key) {
				// p.key == e.dst by the join
			} else if (next.table_index == edge_get->table_index && (dst == INVALID_INDEX || dst == next.column)) {
				dst = next.column;
			} else {
				return nullptr;
			}
		} else if (expr.GetExpressionClass() == ExpressionClass::BOUND_CASE) {
			auto &bound_case = (BoundCaseExpression &)expr;
			if (bound_case.check->type != ExpressionType::COMPARE_EQUAL) {
				return nullptr;
			}
			auto &check = (BoundComparisonExpression &)*bound_case.check;
			PlanColumn l, r, if_true, if_false;
			if (!column_of(*check.left, l) || !column_of(*check.right, r) || !column_of(*bound_case.result_if_true, if_true) ||
			    !column_of(*bound_case.result_if_false, if_false)) {
				return nullptr;
			}
			PlanColumn edge_src;
			edge_src.table_index = edge_get->table_index;
			edge_src.column = src;
			// the check repeats the join condition, so it always holds and the CASE is its THEN branch
			// (RemoveUnusedColumns may already have rewritten e.src into the equal f.friend)
			auto is_join_key = [&](const PlanColumn &c) { return c == cte_friend || c == edge_src; };
			if (!is_join_key(l) || !is_join_key(r) || !is_join_key(if_false) ||
			    if_true.table_index != edge_get->table_index || if_true.column == src ||
			    (dst != INVALID_INDEX && dst != if_true.column)) {
				return nullptr;
			}
			dst = if_true.column;
		} else {
			return nullptr;
		}
	}
	if (dst == INVALID_INDEX || dst == src || src >= edge_table->columns.size() || dst >= edge_table->columns.size() ||
	    !ColumnIsIntegerKey(*edge_table, src) || !ColumnIsIntegerKey(*edge_table, dst)) {
		return nullptr;
	}
	if (validated) {
		if (!ColumnIsUnique(*vertex_table, vertex_key)) {
			return nullptr;
		}
	} else if (!ColumnIsNotNull(*edge_table, dst)) {
		return nullptr; // a NULL destination would surface as a NULL friend on the CPU side
	}
	// ---- output layout of the aggregate: (group 0, group 1, min)
	if (op.types.size() != 3 || op.types[2] != LogicalType::INTEGER) {
		return nullptr;
	}
	vector<unique_ptr<Expression>> select_list;
	for (int g = 0; g < 2; g++) {
		// scan columns: (startPerson BIGINT, friend BIGINT, hopCount INTEGER)
		unique_ptr<Expression> ref =
		    make_unique<BoundReferenceExpression>(LogicalType::BIGINT, group_ref[g] == start_col ? 0 : 1);
		if (op.types[g] != LogicalType::BIGINT) {
			if (op.types[g] != LogicalType::INTEGER) {
				return nullptr;
			}
			ref = make_unique<BoundCastExpression>(move(ref), op.types[g]);
		}
		select_list.push_back(move(ref));
	}
	select_list.push_back(make_unique<BoundReferenceExpression>(LogicalType::INTEGER, 2));

	// ---- the scan
	const string key_name = GGQuote(vertex_table->columns[vertex_key].name);
	GGGraphSpec spec;
	if (validated) {
		spec.vertices = TableColumns(vertex_table, {vertex_key});
	}
	spec.edges = TableColumns(edge_table, {src, dst});
	g_plan_tables.push_back(vertex_table); // the seed rows are read from it by key
	// the seeds are rows of the vertex table (constants that are nobody's key start nothing), read in the
	// statement's own transaction like the edge table: the key column is scanned through the ingest path and
	// the pushed-down filters and the IN list are applied to it here
	if (!all_vertices) {
		string in_list = key_name + " IN (";
		for (idx_t i = 0; i < seed_constants.size(); i++) {
			in_list += (i ? ", " : "") + to_string(seed_constants[i]);
		}
		seed_predicates.push_back(in_list + ")"); // (for EXPLAIN only)
	}
	const auto seed_scan = TableColumns(vertex_table, {vertex_key});
	const int max_hops = (int)in.max_hops;
	const bool lone_sources = !validated;
	auto read_seeds = [=](ClientContext &context) {
		auto sources = GGScanInt64Column(context, seed_scan);
		if (!all_vertices || !seed_filters.empty()) {
			std::unordered_set<int64_t> wanted(seed_constants.begin(), seed_constants.end());
			idx_t kept = 0;
			for (auto id : sources) {
				bool keep = all_vertices || wanted.count(id);
				for (idx_t f = 0; keep && f < seed_filters.size(); f++) {
					keep = seed_filters[f].Accepts(id);
				}
				if (keep) {
					sources[kept++] = id;
				}
			}
			sources.resize(kept);
		}
		return sources;
	};
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &context, GGOpened &opened) {
		opened.graph = GGBuildGraph(context, spec);
		opened.source = make_unique<PhysicalGGShortestPath>(opened.graph, read_seeds(context), max_hops, 0, lone_sources);
	};
	data->description = edge_table->name + ": " + edge_table->columns[src].name + " -> " + edge_table->columns[dst].name +
	                    "\nmin hops <= " + to_string(max_hops) + "\nvertices: " +
	                    (validated ? vertex_table->name + "." + vertex_table->columns[vertex_key].name
	                               : string("endpoint ids")) +
	                    "\nfrom " +
	                    (seed_predicates.empty() ? "every " + vertex_table->name
	                     : all_vertices          ? vertex_table->name + " where " + seed_predicates[0]
	                                             : to_string(seed_constants.size()) +
	                                          (seed_constants.size() == 1 ? " id" : " ids"));
	data->parallel_result = true;
	vector<LogicalType> types = {LogicalType::BIGINT, LogicalType::BIGINT, LogicalType::INTEGER};
	vector<column_t> column_ids = {0, 1, 2};
	vector<string> names = {"startPerson", "friend", "hopCount"};
	g_rules_fired++;
	unique_ptr<PhysicalOperator> scan;
	if (g_plan_context && GGPipelineSinksAvailable(*g_plan_context, spec)) {
		// the tables reach the device through pipeline sinks (gg_pipeline.cpp); the seeds are read when the graph exists
		scan = GGMakeGraphScan(
		    spec, move(types), "GG_SHORTEST_PATH_BFS", data->description, true,
		    [=](ClientContext &context, shared_ptr<GGGraph> graph) -> unique_ptr<PhysicalOperator> {
			    return make_unique<PhysicalGGShortestPath>(move(graph), read_seeds(context), max_hops, 0, lone_sources);
		    },
		    op.estimated_cardinality);
	} else {
		scan = make_unique<PhysicalTableScan>(move(types), GGScanFunction("gg_shortest_path_bfs"), move(data),
		                                      move(column_ids), move(names), nullptr, op.estimated_cardinality);
	}
	auto projection = make_unique<PhysicalProjection>(op.types, move(select_list), op.estimated_cardinality);
	projection->children.push_back(move(scan));
	return move(projection);
}

//===--------------------------------------------------------------------===//
// Distinct rule: the dedupe above a UNION of 1-hop and 2-hop endpoints of one source
//===--------------------------------------------------------------------===//
// benchmark/ldbc/queries/interactive-complex-3.sql:3-12 (and -5, -6, -9, -11): the friends of a person UNION the
// friends of those friends,
//
//     select dst from e where src = C
//     union
//     select e2.dst from e e1, e e2 where e1.src = C and e1.dst = e2.src [and e2.dst <> X ...]
//
// which the reference plans as PhysicalUnion under a hash aggregate that dedupes the ids
// (plan_distinct.cpp:12-78, physical_hash_aggregate.cpp:152-266).  On the device both branches are set images
// under the edge relation (gg_walk_endpoints): one row per endpoint with a flag per walk length, so the UNION
// is a filter — in the 1-hop set, or in the 2-hop set and passing the second branch's own predicates on the endpoint —
// and nothing is left to dedupe.
//! Distinct rule 2 (round 4): `SELECT DISTINCT <end vertex> FROM` a walk of h >= 2 edges over one edge table pinned at
//! its first vertex — the dedupe the reference plans as a hash aggregate above the projection of its joins
//! (src/execution/physical_plan/plan_distinct.cpp:12-78) is the SET IMAGE of h hops from the source: gg_walk_endpoints
//! keeps, per vertex, a flag for every walk length that ends there; the rule keeps the vertices whose flag h is set
//! (and that pass the predicates the statement puts on the end vertex).
unique_ptr<PhysicalOperator> PlanDistinctEndpoints(LogicalDistinct &op) {
	if (op.children.size() != 1 || op.types.size() != 1 || op.children[0]->type != LogicalOperatorType::LOGICAL_PROJECTION) {
		return nullptr;
	}
	if (op.distinct_targets.size() > 1 ||
	    (op.distinct_targets.size() == 1 && (op.distinct_targets[0]->type != ExpressionType::BOUND_REF ||
	                                         ((BoundReferenceExpression &)*op.distinct_targets[0]).index != 0))) {
		return nullptr;
	}
	auto &projection_op = *op.children[0];
	if (projection_op.children.size() != 1 || projection_op.expressions.size() != 1 ||
	    projection_op.expressions[0]->type != ExpressionType::BOUND_REF ||
	    projection_op.children[0]->type != LogicalOperatorType::LOGICAL_COMPARISON_JOIN) {
		return nullptr;
	}
	auto &join = *projection_op.children[0];
	PatternInput in;
	WalkPattern pattern;
	if (!CollectJoinTree(join, in) || !SolveWalkPattern(in, pattern) || pattern.hops < 2 || pattern.hops > GG_MAX_HOPS ||
	    pattern.all_sources || pattern.sources.size() != 1 || pattern.vertex_table || !pattern.payload_filters.empty()) {
		return nullptr;
	}
	const idx_t hops = pattern.hops;
	for (auto &entry : pattern.residual) {
		if (entry.first != hops) {
			return nullptr; // a predicate on an inner vertex changes which walks exist
		}
	}
	{
		auto bindings = join.GetColumnBindings();
		const auto index = ((BoundReferenceExpression &)*projection_op.expressions[0]).index;
		LeafColumn column;
		if (index >= bindings.size() || !ResolveLeafColumn(in, bindings[index], column) ||
		    pattern.edge_position[column.leaf] != hops || column.column != pattern.dst_column) {
			return nullptr;
		}
	}
	auto &table = *pattern.edge_table;
	if (!ColumnIsNotNull(table, pattern.dst_column) || table.columns[pattern.dst_column].type != op.types[0]) {
		return nullptr; // a NULL endpoint is a row of the reference's DISTINCT; the vertex set has no NULL
	}
	const auto spec = GraphSpecOf(pattern);
	const auto sources = pattern.sources;
	const int k_max = (int)hops;
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &context, GGOpened &opened) {
		opened.graph = GGBuildGraph(context, spec);
		opened.source = make_unique<PhysicalGGWalkEndpoints>(opened.graph, sources, k_max, 0);
	};
	data->description = table.name + ": " + table.columns[pattern.src_column].name + " -> " +
	                    table.columns[pattern.dst_column].name + "\ndistinct endpoints of " + to_string(hops) +
	                    " hops\nfrom " + to_string(sources[0]);
	auto types = PhysicalGGWalkEndpoints::OutputTypes(k_max);
	g_rules_fired++;
	unique_ptr<PhysicalOperator> scan;
	if (g_plan_context && GGPipelineSinksAvailable(*g_plan_context, spec)) {
		scan = GGMakeGraphScan(
		    spec, move(types), "GG_WALK_ENDPOINTS", data->description, false,
		    [=](ClientContext &, shared_ptr<GGGraph> graph) -> unique_ptr<PhysicalOperator> {
			    return make_unique<PhysicalGGWalkEndpoints>(move(graph), sources, k_max, 0);
		    },
		    op.estimated_cardinality);
	} else {
		vector<column_t> column_ids;
		vector<string> names;
		for (idx_t c = 0; c < types.size(); c++) {
			column_ids.push_back(c);
			names.push_back("c" + to_string(c));
		}
		scan = make_unique<PhysicalTableScan>(move(types), GGScanFunction("gg_walk_endpoints"), move(data),
		                                      move(column_ids), move(names), nullptr, op.estimated_cardinality);
	}
	// scan columns: (id, h1, ..., hk).  keep: hk = 1 AND the statement's predicates on the end vertex
	unique_ptr<Expression> keep = make_unique<BoundComparisonExpression>(
	    ExpressionType::COMPARE_EQUAL, make_unique<BoundReferenceExpression>(LogicalType::BIGINT, hops),
	    make_unique<BoundConstantExpression>(Value::BIGINT(1)));
	if (!pattern.residual.empty()) {
		auto both = make_unique<BoundConjunctionExpression>(ExpressionType::CONJUNCTION_AND);
		both->children.push_back(move(keep));
		for (auto &entry : pattern.residual) {
			both->children.push_back(FilterToExpression(*entry.second, 0));
		}
		keep = move(both);
	}
	vector<unique_ptr<Expression>> predicates;
	predicates.push_back(move(keep));
	auto filter = make_unique<PhysicalFilter>(scan->types, move(predicates), op.estimated_cardinality);
	filter->children.push_back(move(scan));
	vector<unique_ptr<Expression>> select_list;
	unique_ptr<Expression> ref = make_unique<BoundReferenceExpression>(LogicalType::BIGINT, 0);
	if (op.types[0] != LogicalType::BIGINT) {
		ref = make_unique<BoundCastExpression>(move(ref), op.types[0]);
	}
	select_list.push_back(move(ref));
	auto projection = make_unique<PhysicalProjection>(op.types, move(select_list), op.estimated_cardinality);
	projection->children.push_back(move(filter));
	return move(projection);
}

unique_ptr<PhysicalOperator> PlanDistinctUnion(LogicalDistinct &op) {
	if (op.children.size() == 1 && op.children[0]->type == LogicalOperatorType::LOGICAL_PROJECTION) {
		return PlanDistinctEndpoints(op);
	}
	if (op.children.size() != 1 || op.types.size() != 1 || op.children[0]->type != LogicalOperatorType::LOGICAL_UNION) {
		return nullptr;
	}
	if (op.distinct_targets.size() > 1 ||
	    (op.distinct_targets.size() == 1 && (op.distinct_targets[0]->type != ExpressionType::BOUND_REF ||
	                                         ((BoundReferenceExpression &)*op.distinct_targets[0]).index != 0))) {
		return nullptr;
	}
	auto &setop = *op.children[0];
	if (setop.children.size() != 2 || setop.types.size() != 1) {
		return nullptr;
	}
	// which branch is the join?
	LogicalOperator *one = nullptr, *two = nullptr;
	for (auto &child : setop.children) {
		if (child->type != LogicalOperatorType::LOGICAL_PROJECTION || child->children.size() != 1 ||
		    child->expressions.size() != 1 || child->expressions[0]->type != ExpressionType::BOUND_REF) {
			return nullptr;
		}
		(child->children[0]->type == LogicalOperatorType::LOGICAL_COMPARISON_JOIN ? two : one) = child.get();
	}
	if (!one || !two) {
		return nullptr;
	}
	// the 2-hop branch: a walk of two edges over one edge table, pinned at its first vertex, endpoint projected
	PatternInput in2;
	WalkPattern pattern;
	auto &join = *two->children[0];
	if (!CollectJoinTree(join, in2) || !SolveWalkPattern(in2, pattern) || pattern.hops != 2 || pattern.all_sources ||
	    pattern.sources.size() != 1 || pattern.vertex_table || !pattern.payload_filters.empty()) {
		return nullptr;
	}
	for (auto &entry : pattern.residual) {
		if (entry.first != 2) {
			return nullptr; // a predicate on the middle vertex changes which walks exist
		}
	}
	{
		auto bindings = join.GetColumnBindings();
		const auto index = ((BoundReferenceExpression &)*two->expressions[0]).index;
		LeafColumn column;
		if (index >= bindings.size() || !ResolveLeafColumn(in2, bindings[index], column) ||
		    pattern.edge_position[column.leaf] != 2 || column.column != pattern.dst_column) {
			return nullptr;
		}
	}
	// the 1-hop branch: the same table, source and endpoint columns, the same constant, nothing else
	PatternInput in1;
	if (!CollectJoinTree(*one, in1) || in1.leaves.size() != 1 || in1.leaves[0].table != pattern.edge_table ||
	    !in1.filters.empty() || !in1.other_filters.empty() || in1.constants.size() != 1 || in1.constants[0].column.column != pattern.src_column ||
	    in1.constants[0].value != pattern.sources[0] || in1.aliases.empty()) {
		return nullptr;
	}
	{
		auto &outputs = in1.aliases.back().second; // the projection on top of the branch
		if (in1.aliases.back().first != ((LogicalProjection &)*one).table_index || outputs.size() != 1 ||
		    outputs[0].column != pattern.dst_column) {
			return nullptr;
		}
	}
	auto &table = *pattern.edge_table;
	if (!ColumnIsNotNull(table, pattern.dst_column) || table.columns[pattern.dst_column].type != op.types[0]) {
		return nullptr; // a NULL endpoint is a row of the reference's UNION; the vertex set has no NULL
	}

	const auto spec = GraphSpecOf(pattern);
	const auto sources = pattern.sources;
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &context, GGOpened &opened) {
		opened.graph = GGBuildGraph(context, spec);
		opened.source = make_unique<PhysicalGGWalkEndpoints>(opened.graph, sources, 2, 0);
	};
	data->description = table.name + ": " + table.columns[pattern.src_column].name + " -> " +
	                    table.columns[pattern.dst_column].name + "\ndistinct endpoints of 1..2 hops\nfrom " +
	                    to_string(sources[0]);
	auto types = PhysicalGGWalkEndpoints::OutputTypes(2);
	g_rules_fired++;
	unique_ptr<PhysicalOperator> scan;
	if (g_plan_context && GGPipelineSinksAvailable(*g_plan_context, spec)) {
		scan = GGMakeGraphScan(
		    spec, move(types), "GG_WALK_ENDPOINTS", data->description, false,
		    [=](ClientContext &, shared_ptr<GGGraph> graph) -> unique_ptr<PhysicalOperator> {
			    return make_unique<PhysicalGGWalkEndpoints>(move(graph), sources, 2, 0);
		    },
		    op.estimated_cardinality);
	} else {
		vector<column_t> column_ids;
		vector<string> names;
		for (idx_t c = 0; c < types.size(); c++) {
			column_ids.push_back(c);
			names.push_back("c" + to_string(c));
		}
		scan = make_unique<PhysicalTableScan>(move(types), GGScanFunction("gg_walk_endpoints"), move(data),
		                                      move(column_ids), move(names), nullptr, op.estimated_cardinality);
	}
	// scan columns: (id, h1, h2).  keep: h1 = 1 OR (h2 = 1 AND the second branch's predicates on the endpoint)
	auto flag = [](idx_t column) {
		return make_unique<BoundComparisonExpression>(ExpressionType::COMPARE_EQUAL,
		                                              make_unique<BoundReferenceExpression>(LogicalType::BIGINT, column),
		                                              make_unique<BoundConstantExpression>(Value::BIGINT(1)));
	};
	unique_ptr<Expression> second = flag(2);
	if (!pattern.residual.empty()) {
		auto both = make_unique<BoundConjunctionExpression>(ExpressionType::CONJUNCTION_AND);
		both->children.push_back(move(second));
		for (auto &entry : pattern.residual) {
			both->children.push_back(FilterToExpression(*entry.second, 0));
		}
		second = move(both);
	}
	auto keep = make_unique<BoundConjunctionExpression>(ExpressionType::CONJUNCTION_OR);
	keep->children.push_back(flag(1));
	keep->children.push_back(move(second));
	vector<unique_ptr<Expression>> predicates;
	predicates.push_back(move(keep));
	auto filter = make_unique<PhysicalFilter>(scan->types, move(predicates), op.estimated_cardinality);
	filter->children.push_back(move(scan));
	vector<unique_ptr<Expression>> select_list;
	unique_ptr<Expression> ref = make_unique<BoundReferenceExpression>(LogicalType::BIGINT, 0);
	if (op.types[0] != LogicalType::BIGINT) {
		ref = make_unique<BoundCastExpression>(move(ref), op.types[0]);
	}
	select_list.push_back(move(ref));
	auto projection = make_unique<PhysicalProjection>(op.types, move(select_list), op.estimated_cardinality);
	projection->children.push_back(move(filter));
	return move(projection);
}

unique_ptr<PhysicalOperator> PlanAggregate(LogicalAggregate &op) {
	if (auto plan = PlanCountOverJoinChain(op)) {
		return plan;
	}
	return PlanShortestPath(op);
}

//! The rule RULE offered the logical operator, for the connection `context` (dependencies and rec_ctes are public
//! members of the generator).
template <class OP, unique_ptr<PhysicalOperator> (*RULE)(OP &)>
int RuleEntryFor(ClientContext &context, void *ret_slot, void *generator, void *logical_operator) {
	if (!GGGetConnectionFlags(context).rules) {
		return 0; // PRAGMA enable_gpu_graph was not issued on THIS connection
	}
	unique_ptr<PhysicalOperator> plan;
	g_plan_tables.clear();
	g_plan_context = &context;
	g_plan_generator = (PhysicalPlanGenerator *)generator;
	try {
		plan = RULE(*(OP *)logical_operator);
	} catch (std::exception &) {
		return 0; // whatever went wrong while looking: the reference's planner takes over
	}
	if (!plan) {
		return 0;
	}
	for (auto table : g_plan_tables) {
		((PhysicalPlanGenerator *)generator)->dependencies.insert(table);
	}
	// planned inside a statement with a recursive CTE (the generator registers the working table before it plans the
	// CTE's arms, plan_recursive_cte.cpp:14-21): the pipelines of a recursive arm are reset and re-run per iteration,
	// so a graph scan in there keeps its graph across its source states
	if (!((PhysicalPlanGenerator *)generator)->rec_ctes.empty()) {
		GGKeepGraphs(*plan);
	}
	new (ret_slot) unique_ptr<PhysicalOperator>(move(plan));
	return 1;
}

#ifndef GG_REFERENCE_CALLOUTS
//! Behind the interposition shim (gg_plan_hook.c): the generator is all there is, its connection read past `private`
template <class OP, unique_ptr<PhysicalOperator> (*RULE)(OP &)>
int RuleEntry(void *ret_slot, void *generator, void *logical_operator) {
	return RuleEntryFor<OP, RULE>(((PhysicalPlanGenerator *)generator)->context, ret_slot, generator, logical_operator);
}
#endif

//! The same rules behind the call-outs of a patched reference (oracle/callout.patch):
//! unique_ptr<PhysicalOperator> (*)(ClientContext &, PhysicalPlanGenerator &, LogicalOperator &), null = not taken over
template <class OP, unique_ptr<PhysicalOperator> (*RULE)(OP &)>
unique_ptr<PhysicalOperator> CalloutEntry(ClientContext &context, PhysicalPlanGenerator &generator, LogicalOperator &op) {
	typename std::aligned_storage<sizeof(unique_ptr<PhysicalOperator>), alignof(unique_ptr<PhysicalOperator>)>::type slot;
	if (!RuleEntryFor<OP, RULE>(context, &slot, &generator, &op)) {
		return nullptr;
	}
	auto made = reinterpret_cast<unique_ptr<PhysicalOperator> *>(&slot);
	auto plan = move(*made);
	made->~unique_ptr<PhysicalOperator>();
	return plan;
}

void PragmaEnableGpuGraph(ClientContext &context, const FunctionParameters &parameters) {
	auto flags = GGGetConnectionFlags(context);
	flags.rules = true;
	GGSetConnectionFlags(context, flags);
}

void PragmaDisableGpuGraph(ClientContext &context, const FunctionParameters &parameters) {
	auto flags = GGGetConnectionFlags(context);
	flags.rules = false;
	GGSetConnectionFlags(context, flags);
}

void PragmaEnableGpuJoins(ClientContext &context, const FunctionParameters &parameters) {
	auto flags = GGGetConnectionFlags(context);
	flags.joins = true;
	GGSetConnectionFlags(context, flags);
}

void PragmaDisableGpuJoins(ClientContext &context, const FunctionParameters &parameters) {
	auto flags = GGGetConnectionFlags(context);
	flags.joins = false;
	GGSetConnectionFlags(context, flags);
}

void PragmaUsePinnedGraphs(ClientContext &context, const FunctionParameters &parameters) {
	auto flags = GGGetConnectionFlags(context);
	flags.pinned_graphs = true;
	GGSetConnectionFlags(context, flags);
}

void PragmaIgnorePinnedGraphs(ClientContext &context, const FunctionParameters &parameters) {
	auto flags = GGGetConnectionFlags(context);
	flags.pinned_graphs = false;
	GGSetConnectionFlags(context, flags);
}

//! INSERT / DELETE / UPDATE about to be planned: drop the graphs pinned on the target table (never takes the
//! plan over)
template <class OP>
int WriteObserver(void *ret_slot, void *generator, void *logical_operator) {
	auto table = ((OP *)logical_operator)->table;
	if (table) {
		GGDropPinsOfTable(table->oid);
	}
	return 0;
}

} // namespace

void GGRegisterPlanRules(ClientContext &context) {
	// same style as the reference's enable_profiling / disable_profiling (pragma_functions.cpp:280-345)
	CreatePragmaFunctionInfo enable(PragmaFunction::PragmaStatement("enable_gpu_graph", PragmaEnableGpuGraph));
	CreatePragmaFunctionInfo disable(PragmaFunction::PragmaStatement("disable_gpu_graph", PragmaDisableGpuGraph));
	CreatePragmaFunctionInfo use_pins(PragmaFunction::PragmaStatement("gg_use_pinned_graphs", PragmaUsePinnedGraphs));
	CreatePragmaFunctionInfo no_pins(
	    PragmaFunction::PragmaStatement("gg_ignore_pinned_graphs", PragmaIgnorePinnedGraphs));
	Catalog::GetCatalog(context).CreatePragmaFunction(context, &enable);
	Catalog::GetCatalog(context).CreatePragmaFunction(context, &disable);
	Catalog::GetCatalog(context).CreatePragmaFunction(context, &use_pins);
	Catalog::GetCatalog(context).CreatePragmaFunction(context, &no_pins);
	CreatePragmaFunctionInfo joins_on(PragmaFunction::PragmaStatement("enable_gpu_joins", PragmaEnableGpuJoins));
	CreatePragmaFunctionInfo joins_off(PragmaFunction::PragmaStatement("disable_gpu_joins", PragmaDisableGpuJoins));
	Catalog::GetCatalog(context).CreatePragmaFunction(context, &joins_on);
	Catalog::GetCatalog(context).CreatePragmaFunction(context, &joins_off);

	// A reference built with oracle/callout.patch exports the registration of its call-outs: the maintainers' route —
	// no interposition, no access to private members (the BuildPipelines case and the write observation are then the
	// executor's own code, src/parallel/executor.cpp as patched)
	using plan_fn = unique_ptr<PhysicalOperator> (*)(ClientContext &, PhysicalPlanGenerator &, LogicalOperator &);
	using write_fn = void (*)(idx_t);
	auto callouts = (void (*)(plan_fn, plan_fn, plan_fn, write_fn))dlsym(RTLD_DEFAULT, "duckdb_register_plan_callouts");
	if (callouts) {
		callouts(CalloutEntry<LogicalComparisonJoin, PlanJoin>, CalloutEntry<LogicalAggregate, PlanAggregate>,
		         CalloutEntry<LogicalDistinct, PlanDistinctUnion>, GGDropPinsOfTable);
		g_callouts_registered = true;
		GGPipelineSinksNative();
		return;
	}
#ifndef GG_REFERENCE_CALLOUTS
	// otherwise the interposition shim, if it was loaded before libduckdb; without either the extension only offers
	// its table functions
	auto reg = (int (*)(int, gg_plan_rule_fn))dlsym(RTLD_DEFAULT, "gg_plan_hook_register");
	if (!reg) {
		return;
	}
	reg(GG_PLAN_HOOK_JOIN, RuleEntry<LogicalComparisonJoin, PlanJoin>);
	reg(GG_PLAN_HOOK_AGGREGATE, RuleEntry<LogicalAggregate, PlanAggregate>);
	reg(GG_PLAN_HOOK_DISTINCT, RuleEntry<LogicalDistinct, PlanDistinctUnion>);
	GGRegisterPipelineRule();
	reg(GG_PLAN_HOOK_INSERT, WriteObserver<LogicalInsert>);
	reg(GG_PLAN_HOOK_DELETE, WriteObserver<LogicalDelete>);
	reg(GG_PLAN_HOOK_UPDATE, WriteObserver<LogicalUpdate>);
	// GG_PLAN_RULE=1 in the environment is the default of connections that issue no pragma (gg_duckdb_extension.cpp)
#endif
}

} // namespace duckdb

extern "C" {
//! 1 when the rules are registered: with the call-outs of a patched reference, or with the loaded shim
int gg_plan_rules_available() {
	if (duckdb::g_callouts_registered) {
		return 1;
	}
	auto probe = (int (*)(int))dlsym(RTLD_DEFAULT, "gg_plan_hook_registered");
	return probe && probe(GG_PLAN_HOOK_JOIN) ? 1 : 0;
}
//! 1 when they are registered through the call-outs of a patched reference (no interposition)
int gg_plan_rules_by_callout() {
	return duckdb::g_callouts_registered ? 1 : 0;
}
//! number of plans a rule took over since load
uint64_t gg_plan_rules_fired() {
	return duckdb::g_rules_fired.load();
}
}
