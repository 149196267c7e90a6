/* gg_plan_hook.h — contract between the interposition shim (gg_plan_hook.c, libgg_plan_hook.so) and the
 * planner rule of the extension (gg_plan_rule.cpp). */
#ifndef GG_PLAN_HOOK_H
#define GG_PLAN_HOOK_H
#ifdef __cplusplus
extern "C" {
#endif

enum {
  GG_PLAN_HOOK_JOIN = 0,
  GG_PLAN_HOOK_AGGREGATE = 1,
  /* INSERT / DELETE / UPDATE plans are only OBSERVED (the rule returns 0): a statement that is about to change a
   * table drops the graphs pinned on it (gg_duckdb_extension.cpp, Pinned graphs) */
  GG_PLAN_HOOK_INSERT = 2,
  GG_PLAN_HOOK_DELETE = 3,
  GG_PLAN_HOOK_UPDATE = 4,
  /* DISTINCT (the dedupe above a UNION of walk endpoints) */
  GG_PLAN_HOOK_DISTINCT = 5,
  /* not a CreatePlan: Executor::BuildPipelines(PhysicalOperator *, Pipeline *) — the rule gets (executor, operator,
   * current pipeline) and returns non-zero if it built the operator's pipelines itself (gg_pipeline.cpp) */
  GG_PLAN_HOOK_PIPELINES = 6,
  GG_PLAN_HOOK_KINDS = 7
};

/* A rule looks at the logical operator about to be planned.  To take it over it constructs a
 * std::unique_ptr<duckdb::PhysicalOperator> in *ret_slot (placement new) and returns non-zero; returning
 * zero leaves ret_slot untouched and the reference's own CreatePlan runs. */
typedef int (*gg_plan_rule_fn)(void *ret_slot, void *physical_plan_generator, void *logical_operator);

int gg_plan_hook_register(int kind, gg_plan_rule_fn rule);
int gg_plan_hook_registered(int kind);
/* number of hooks this build of the shim knows (an extension newer than the shim checks before registering) */
int gg_plan_hook_kinds(void);
/* libduckdb's own definition of the function hook `kind` interposes (NULL if none is loaded) */
void *gg_plan_hook_original(int kind);

#ifdef __cplusplus
}
#endif
#endif
