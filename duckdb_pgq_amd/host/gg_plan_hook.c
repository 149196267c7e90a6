/* gg_plan_hook.c — plan-level substitution without touching the reference's sources.
 *
 * The maintainers' route to let MATCH-shaped plans use the GPU operators is a three-line call-out in
 * PhysicalPlanGenerator::CreatePlan(LogicalComparisonJoin &) / (LogicalAggregate &)
 * (src/execution/physical_plan/plan_comparison_join.cpp:146, plan_aggregate.cpp:20 of the reference;
 * INTEGRATION.md §3).  The reference tree is read-only here, so this shim supplies the same call-out from
 * the outside: it DEFINES those two member functions under their Itanium-mangled names.  Loaded before
 * libduckdb (LD_PRELOAD, or dlopen(RTLD_GLOBAL) from the host process), the dynamic linker binds
 * libduckdb's own PLT calls to the definitions below; each one offers the logical operator to the rule
 * the extension registered (gg_plan_rule.cpp) and otherwise forwards to libduckdb's original.
 *
 * Deliberately plain C with no duckdb header: the shim must load before libduckdb does, so it cannot have
 * undefined references to its vtables.  The calling convention is spelled out instead (x86-64 SysV):
 * a function returning std::unique_ptr<PhysicalOperator> takes the address of the return slot as a hidden
 * first argument and returns that address; `this` and the operator reference follow.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <link.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "gg_plan_hook.h"

#define SYM_JOIN "_ZN6duckdb21PhysicalPlanGenerator10CreatePlanERNS_21LogicalComparisonJoinE"
#define SYM_AGGR "_ZN6duckdb21PhysicalPlanGenerator10CreatePlanERNS_16LogicalAggregateE"
#define SYM_INSERT "_ZN6duckdb21PhysicalPlanGenerator10CreatePlanERNS_13LogicalInsertE"
#define SYM_DELETE "_ZN6duckdb21PhysicalPlanGenerator10CreatePlanERNS_13LogicalDeleteE"
#define SYM_UPDATE "_ZN6duckdb21PhysicalPlanGenerator10CreatePlanERNS_13LogicalUpdateE"
#define SYM_DISTINCT "_ZN6duckdb21PhysicalPlanGenerator10CreatePlanERNS_15LogicalDistinctE"
#define SYM_PIPELINES "_ZN6duckdb8Executor14BuildPipelinesEPNS_16PhysicalOperatorEPNS_8PipelineE"

typedef void *(*create_plan_fn)(void *ret_slot, void *generator, void *logical_op);

static gg_plan_rule_fn g_rule[GG_PLAN_HOOK_KINDS];
static create_plan_fn g_orig[GG_PLAN_HOOK_KINDS];
static const char *const g_sym[GG_PLAN_HOOK_KINDS] = {SYM_JOIN, SYM_AGGR, SYM_INSERT, SYM_DELETE, SYM_UPDATE,
                                                          SYM_DISTINCT, SYM_PIPELINES};

void *gg_hook_create_plan_join(void *ret_slot, void *generator, void *op) __asm__(SYM_JOIN);
void *gg_hook_create_plan_aggregate(void *ret_slot, void *generator, void *op) __asm__(SYM_AGGR);
void *gg_hook_create_plan_insert(void *ret_slot, void *generator, void *op) __asm__(SYM_INSERT);
void *gg_hook_create_plan_delete(void *ret_slot, void *generator, void *op) __asm__(SYM_DELETE);
void *gg_hook_create_plan_update(void *ret_slot, void *generator, void *op) __asm__(SYM_UPDATE);
void *gg_hook_create_plan_distinct(void *ret_slot, void *generator, void *op) __asm__(SYM_DISTINCT);
/* void Executor::BuildPipelines(PhysicalOperator *op, Pipeline *current): `this`, then the two pointers; no return slot */
void gg_hook_build_pipelines(void *executor, void *op, void *current) __asm__(SYM_PIPELINES);

struct find_ctx {
  const char *sym;
  void *self;
  void *found;
};

/* walk the loaded objects for another definition of `sym` (libduckdb may have been dlopen'ed RTLD_LOCAL,
 * where RTLD_NEXT does not reach it) */
static int find_in_object(struct dl_phdr_info *info, size_t size, void *data) {
  struct find_ctx *c = (struct find_ctx *)data;
  (void)size;
  if (!info->dlpi_name || !info->dlpi_name[0]) return 0;
  void *h = dlopen(info->dlpi_name, RTLD_LAZY | RTLD_NOLOAD);
  if (!h) return 0;
  void *p = dlsym(h, c->sym);
  dlclose(h);
  if (p && p != c->self) {
    c->found = p;
    return 1;
  }
  return 0;
}

static create_plan_fn original(int kind, void *self) {
  if (g_orig[kind]) return g_orig[kind];
  void *p = dlsym(RTLD_NEXT, g_sym[kind]);
  if (!p || p == self) {
    struct find_ctx c = {g_sym[kind], self, NULL};
    dl_iterate_phdr(find_in_object, &c);
    p = c.found;
  }
  if (!p) {
    fprintf(stderr, "gg_plan_hook: no original definition of %s is loaded\n", g_sym[kind]);
    return NULL;
  }
  g_orig[kind] = (create_plan_fn)p;
  return g_orig[kind];
}

static void *dispatch(int kind, void *self, void *ret_slot, void *generator, void *op) {
  gg_plan_rule_fn rule = g_rule[kind];
  if (rule && rule(ret_slot, generator, op)) return ret_slot; /* the rule constructed the plan in place */
  create_plan_fn orig = original(kind, self);
  if (!orig) {
    *(void **)ret_slot = NULL; /* empty unique_ptr: the caller will fail on it rather than run wild */
    return ret_slot;
  }
  return orig(ret_slot, generator, op);
}

void *gg_hook_create_plan_join(void *ret_slot, void *generator, void *op) {
  return dispatch(GG_PLAN_HOOK_JOIN, (void *)gg_hook_create_plan_join, ret_slot, generator, op);
}

void *gg_hook_create_plan_aggregate(void *ret_slot, void *generator, void *op) {
  return dispatch(GG_PLAN_HOOK_AGGREGATE, (void *)gg_hook_create_plan_aggregate, ret_slot, generator, op);
}

void *gg_hook_create_plan_insert(void *ret_slot, void *generator, void *op) {
  return dispatch(GG_PLAN_HOOK_INSERT, (void *)gg_hook_create_plan_insert, ret_slot, generator, op);
}

void *gg_hook_create_plan_delete(void *ret_slot, void *generator, void *op) {
  return dispatch(GG_PLAN_HOOK_DELETE, (void *)gg_hook_create_plan_delete, ret_slot, generator, op);
}

void *gg_hook_create_plan_update(void *ret_slot, void *generator, void *op) {
  return dispatch(GG_PLAN_HOOK_UPDATE, (void *)gg_hook_create_plan_update, ret_slot, generator, op);
}

void *gg_hook_create_plan_distinct(void *ret_slot, void *generator, void *op) {
  return dispatch(GG_PLAN_HOOK_DISTINCT, (void *)gg_hook_create_plan_distinct, ret_slot, generator, op);
}

void gg_hook_build_pipelines(void *executor, void *op, void *current) {
  /* Only the call from Executor::Initialize arrives here: libduckdb's recursive calls of BuildPipelines are direct.
   * The rule therefore wraps the whole traversal: it calls the original itself (gg_plan_hook_original). */
  gg_plan_rule_fn rule = g_rule[GG_PLAN_HOOK_PIPELINES];
  if (rule && rule(executor, op, current)) return;
  create_plan_fn orig = original(GG_PLAN_HOOK_PIPELINES, (void *)gg_hook_build_pipelines);
  if (orig) orig(executor, op, current); /* (same three pointer arguments; nothing is returned) */
}

void *gg_plan_hook_original(int kind) {
  static void *const self[GG_PLAN_HOOK_KINDS] = {
      (void *)gg_hook_create_plan_join,   (void *)gg_hook_create_plan_aggregate, (void *)gg_hook_create_plan_insert,
      (void *)gg_hook_create_plan_delete, (void *)gg_hook_create_plan_update,    (void *)gg_hook_create_plan_distinct,
      (void *)gg_hook_build_pipelines};
  if (kind < 0 || kind >= GG_PLAN_HOOK_KINDS) return NULL;
  return (void *)original(kind, self[kind]);
}

int gg_plan_hook_kinds(void) {
  return GG_PLAN_HOOK_KINDS;
}

int gg_plan_hook_register(int kind, gg_plan_rule_fn rule) {
  if (kind < 0 || kind >= GG_PLAN_HOOK_KINDS) return -1;
  g_rule[kind] = rule;
  return 0;
}

int gg_plan_hook_registered(int kind) {
  return kind >= 0 && kind < GG_PLAN_HOOK_KINDS && g_rule[kind] != NULL;
}
