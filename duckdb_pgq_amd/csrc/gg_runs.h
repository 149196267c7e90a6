// Runs of one key inside a wave: LDS counter updates for edge tables sorted by an endpoint.
// Used by the CSR build kernels (gg_csr_fast.hip); scripts/ubench_runadd.hip checks it lane by lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gg {

// ---- runs of one key inside a wave (edge tables sorted by an endpoint) -------------------------------------------
// Lanes that hold the same key serialise on its LDS counter: ds_add on one address retires a lane per cycle or so,
// and a table sorted by source id (as LDBC ships `knows`) puts ~45 equal keys side by side in every kernel of the
// forward direction (measured at SF100: partition 197 -> 340 us, sub sort 125 -> 334, leaves 354 -> 503).  With
// runs, only the FIRST lane of each run of equal keys adds (the run's length) and the others take their slot from it.
// Order: a run's lanes get consecutive values in lane order, and the first lanes of several runs of one key are
// served in lane order like any other lanes (lds_order_ok), so the ranks stay stable.
// x of lane - 1 (lane 0: its own).  Call with the whole wave active.
__device__ __forceinline__ uint32_t lane_before(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
// does this step of the wave consist of runs?  (a quarter of the lanes or fewer start one)
#ifndef GG_FB_RUNAGG
#define GG_FB_RUNAGG 1
#endif
__device__ __forceinline__ bool wave_has_runs(uint32_t key, bool valid, int lane) {
  if (!GG_FB_RUNAGG) return false;
  const uint32_t kk = valid ? key : 0xFFFFFFFFu;
  const uint32_t pk = lane_before(kk);  // (every lane executes the move: a masked-off lane cannot be read)
  return __popcll(__ballot(lane == 0 || pk != kk)) <= 16;
}
// valid lanes: *(cur + key) += 1; returns the value before the lane's own increment (RTN) or nothing
template <bool RTN>
__device__ __forceinline__ uint32_t run_add(uint32_t *cur, uint32_t key, bool valid, int lane) {
  const uint32_t kk = valid ? key : 0xFFFFFFFFu;
  const uint32_t pk = lane_before(kk);  // (every lane executes the move: a masked-off lane cannot be read)
  const bool lead = lane == 0 || pk != kk;
  const uint64_t lm = __ballot(lead);
  const uint64_t upto = (2ULL << lane) - 1ULL;  // lanes 0..lane
  const uint64_t after = lm & ~upto;
  const int end = after ? __ffsll((unsigned long long)after) - 1 : 64;  // the next run starts here
  uint32_t base = 0;
  if (lead && valid) {
    if (RTN) {
      base = atomicAdd(cur + key, (uint32_t)(end - lane));
    } else {
      atomicAdd(cur + key, (uint32_t)(end - lane));
    }
  }
  if (!RTN) return 0;
  const int first = 63 - __clzll((unsigned long long)(lm & upto));  // first lane of this lane's run
  return (uint32_t)__shfl((int)base, first, 64) + (uint32_t)(lane - first);
}

}  // namespace gg
