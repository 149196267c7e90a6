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
//                   is materialised), one output row.
//
// A pattern is accepted only when the substitution is exact for every database state:
//   * every leaf is a plain sequential scan, every join INNER with only column = column conditions; the
//     only pushed-down filter allowed is `= constant` on the walk's first vertex (a single source);
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
#include "duckdb/execution/operator/projection/physical_projection.hpp"
#include "duckdb/execution/operator/scan/physical_table_scan.hpp"
#include "duckdb/execution/physical_plan_generator.hpp"
#include "duckdb/function/pragma_function.hpp"
#include "duckdb/function/table/table_scan.hpp"
#include "duckdb/parser/parsed_data/create_pragma_function_info.hpp"
#include "duckdb/planner/constraints/bound_not_null_constraint.hpp"
#include "duckdb/planner/constraints/bound_unique_constraint.hpp"
#include "duckdb/planner/expression/bound_aggregate_expression.hpp"
#include "duckdb/planner/expression/bound_cast_expression.hpp"
#include "duckdb/planner/expression/bound_reference_expression.hpp"
#include "duckdb/planner/filter/conjunction_filter.hpp"
#include "duckdb/planner/filter/constant_filter.hpp"
#include "duckdb/planner/operator/logical_aggregate.hpp"
#include "duckdb/planner/operator/logical_comparison_join.hpp"
#include "duckdb/planner/operator/logical_get.hpp"
#include "gg_extension.hpp"
#include "gg_plan_hook.h"

namespace duckdb {

static std::atomic<bool> g_rules_enabled {false};
static std::atomic<uint64_t> g_rules_fired {0};

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
};

struct PatternInput {
	vector<ScanLeaf> leaves;
	vector<std::pair<LeafColumn, LeafColumn>> equalities;
	vector<LeafConstant> constants;
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
	//! per leaf: walk position of its src column (edge leaves: position of dst is +1) or of its key
	vector<idx_t> edge_position;   // leaf -> 1-based edge number, 0 for vertex leaves
	vector<idx_t> vertex_position; // leaf -> walk position (vertex leaves only)
};

bool ResolveLeafColumn(const PatternInput &in, const ColumnBinding &binding, LeafColumn &out) {
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
			if (entry.first >= bind.table->columns.size() || !ColumnIsIntegerKey(*bind.table, entry.first) ||
			    !IsEqualityWithConstant(*entry.second, value)) {
				return false;
			}
			in.constants.push_back({{in.leaves.size() - 1, entry.first}, value});
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
	// pinned constants: all on walk position 0 and all the same value (the optimizer copies a constant
	// to every column it is transitively equal to)
	out.all_sources = in.constants.empty();
	out.sources.clear();
	for (auto &constant : in.constants) {
		const auto l = constant.column.leaf;
		const bool at_start = (out.edge_position[l] == 1 && constant.column.column == src) ||
		                      (out.edge_position[l] == 0 && out.vertex_position[l] == 0 &&
		                       constant.column.column == vertex_key);
		if (!at_start || (!out.sources.empty() && out.sources[0] != constant.value)) {
			return false;
		}
		out.sources.assign(1, constant.value);
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
		if (SolveWithRoles(in, classes, edge_table, vertex_table, used[0], used[1], out) ||
		    SolveWithRoles(in, classes, edge_table, vertex_table, used[1], used[0], out)) {
			return true;
		}
	}
	return false;
}

string QualifiedName(TableCatalogEntry &table) {
	return GGQuote(table.schema->name) + "." + GGQuote(table.name);
}

GGGraphSpec GraphSpecOf(const WalkPattern &pattern) {
	GGGraphSpec spec;
	if (pattern.vertex_table) {
		spec.vertex_sql = "SELECT " + GGQuote(pattern.vertex_table->columns[pattern.vertex_key].name) + " FROM " +
		                  QualifiedName(*pattern.vertex_table);
	}
	spec.edge_sql = "SELECT " + GGQuote(pattern.edge_table->columns[pattern.src_column].name) + ", " +
	                GGQuote(pattern.edge_table->columns[pattern.dst_column].name) + " FROM " +
	                QualifiedName(*pattern.edge_table);
	return spec;
}

//! PhysicalTableScan over the gg scan function for `hops`-hop walks of the pattern.
unique_ptr<PhysicalOperator> MakeExpandScan(const WalkPattern &pattern, bool count_only, idx_t estimated_cardinality) {
	const auto spec = GraphSpecOf(pattern);
	const int hops = (int)pattern.hops;
	const auto sources = pattern.sources;
	const bool all_sources = pattern.all_sources;
	auto data = make_unique<GGFunctionData>();
	data->open = [=](ClientContext &context, GGOpened &opened) {
		opened.graph = GGBuildGraph(context, spec);
		opened.source = make_unique<PhysicalGGPathExpand>(opened.graph, hops, hops, count_only, sources, all_sources, 0);
	};
	data->description = pattern.edge_table->name + ": " + pattern.edge_table->columns[pattern.src_column].name +
	                    " -> " + pattern.edge_table->columns[pattern.dst_column].name + "\n" + to_string(hops) +
	                    (hops == 1 ? " hop" : " hops") + "\nvertices: " +
	                    (pattern.vertex_table ? pattern.vertex_table->name + "." +
	                                                pattern.vertex_table->columns[pattern.vertex_key].name
	                                          : string("endpoint ids")) +
	                    (all_sources ? string() : "\nfrom " + to_string(sources[0]));
	auto types = PhysicalGGPathExpand::OutputTypes(hops, count_only);
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

//! Join rule: the subtree's columns are walk vertices.
unique_ptr<PhysicalOperator> PlanJoinChain(LogicalComparisonJoin &op) {
	PatternInput in;
	WalkPattern pattern;
	if (!CollectJoinTree(op, in) || !SolveWalkPattern(in, pattern)) {
		return nullptr;
	}
	// output layout of the join: map every column to a walk position
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
		idx_t position;
		auto &table = *in.leaves[column.leaf].table;
		if (pattern.edge_position[column.leaf]) {
			if (column.column == pattern.src_column) {
				position = pattern.edge_position[column.leaf] - 1;
			} else if (column.column == pattern.dst_column) {
				position = pattern.edge_position[column.leaf];
			} else {
				return nullptr;
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
	auto scan = MakeExpandScan(pattern, false, op.estimated_cardinality);
	auto projection = make_unique<PhysicalProjection>(op.types, move(select_list), op.estimated_cardinality);
	projection->children.push_back(move(scan));
	return move(projection);
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
	if (!CollectJoinTree(*child, in) || !SolveWalkPattern(in, pattern)) {
		return nullptr;
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
	auto scan = MakeExpandScan(pattern, true, 1);
	auto projection = make_unique<PhysicalProjection>(op.types, move(select_list), 1);
	projection->children.push_back(move(scan));
	return move(projection);
}

template <class OP, unique_ptr<PhysicalOperator> (*RULE)(OP &)>
int RuleEntry(void *ret_slot, void *generator, void *logical_operator) {
	if (!g_rules_enabled) {
		return 0;
	}
	unique_ptr<PhysicalOperator> plan;
	try {
		plan = RULE(*(OP *)logical_operator);
	} catch (std::exception &) {
		return 0; // whatever went wrong while looking: the reference's planner takes over
	}
	if (!plan) {
		return 0;
	}
	new (ret_slot) unique_ptr<PhysicalOperator>(move(plan));
	return 1;
}

void PragmaEnableGpuGraph(ClientContext &context, const FunctionParameters &parameters) {
	g_rules_enabled = true;
}

void PragmaDisableGpuGraph(ClientContext &context, const FunctionParameters &parameters) {
	g_rules_enabled = false;
}

} // namespace

void GGRegisterPlanRules(ClientContext &context) {
	// same style as the reference's enable_profiling / disable_profiling (pragma_functions.cpp:280-345)
	CreatePragmaFunctionInfo enable(PragmaFunction::PragmaStatement("enable_gpu_graph", PragmaEnableGpuGraph));
	CreatePragmaFunctionInfo disable(PragmaFunction::PragmaStatement("disable_gpu_graph", PragmaDisableGpuGraph));
	Catalog::GetCatalog(context).CreatePragmaFunction(context, &enable);
	Catalog::GetCatalog(context).CreatePragmaFunction(context, &disable);

	// the shim is optional: without it the extension only offers its table functions
	auto reg = (int (*)(int, gg_plan_rule_fn))dlsym(RTLD_DEFAULT, "gg_plan_hook_register");
	if (!reg) {
		return;
	}
	reg(GG_PLAN_HOOK_JOIN, RuleEntry<LogicalComparisonJoin, PlanJoinChain>);
	reg(GG_PLAN_HOOK_AGGREGATE, RuleEntry<LogicalAggregate, PlanCountOverJoinChain>);
	auto env = std::getenv("GG_PLAN_RULE");
	if (env && env[0] == '1') {
		g_rules_enabled = true;
	}
}

} // namespace duckdb

extern "C" {
//! 1 when the shim is loaded and the rules are registered with it
int gg_plan_rules_available() {
	auto probe = (int (*)(int))dlsym(RTLD_DEFAULT, "gg_plan_hook_registered");
	return probe && probe(GG_PLAN_HOOK_JOIN) ? 1 : 0;
}
//! number of plans a rule took over since load
uint64_t gg_plan_rules_fired() {
	return duckdb::g_rules_fired.load();
}
}
