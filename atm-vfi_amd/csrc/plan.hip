// Launch plans (include/atmvfi.h, "Launch plans"): a recorded sequence of launch entry points replayed by one call.
// No kernel lives here: every op goes through the same extern "C" entry point -- and its host-side validation -- as a direct call.
#include "common.h"

#include <string.h>

#include "plan_dispatch.inc"

extern "C" int atmvfi_plan_fn_id(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < ATMVFI_PLAN_NFN; ++i)
        if (strcmp(name, kPlanFnNames[i]) == 0) return i;
    return -1;
}

static int plan_patch(atmvfi_plan_op* ops, int n_ops, const atmvfi_plan_patch* patches, int n_patches, const uint64_t* slots, int n_slots) {
    ATMVFI_REQUIRE(ops && n_ops > 0, ATMVFI_EINVAL, "plan_run: empty plan");
    ATMVFI_REQUIRE(n_patches == 0 || (patches && slots), ATMVFI_EINVAL, "plan_run: patches without a slot table");
    for (int i = 0; i < n_patches; ++i) {
        const atmvfi_plan_patch& p = patches[i];
        ATMVFI_REQUIRE(p.op >= 0 && p.op < n_ops && p.arg >= 0 && p.arg < ATMVFI_PLAN_MAX_ARGS && p.slot >= 0 && p.slot < n_slots,
                       ATMVFI_EINVAL, "plan_run: patch %d out of range (op %d arg %d slot %d)", i, p.op, p.arg, p.slot);
        ATMVFI_REQUIRE(slots[p.slot] != 0, ATMVFI_EINVAL, "plan_run: slot %d of patch %d is null", p.slot, i);
        ops[p.op].a[p.arg].u = slots[p.slot] + (uint64_t)p.offset;
    }
    return ATMVFI_OK;
}

extern "C" int atmvfi_plan_run(atmvfi_plan_op* ops, int n_ops, const atmvfi_plan_patch* patches, int n_patches, const uint64_t* slots,
                               int n_slots, int* failed_op, void* stream) {
    void* const one[1] = {stream};
    return atmvfi_plan_run_lanes(ops, n_ops, nullptr, patches, n_patches, slots, n_slots, failed_op, one, 1, nullptr, 0);
}

extern "C" int atmvfi_plan_run_lanes(atmvfi_plan_op* ops, int n_ops, const int32_t* lanes, const atmvfi_plan_patch* patches, int n_patches,
                                     const uint64_t* slots, int n_slots, int* failed_op, void* const* streams, int n_streams,
                                     void* const* events, int n_events) {
    if (failed_op) *failed_op = -1;
    ATMVFI_REQUIRE(streams && n_streams >= 1, ATMVFI_EINVAL, "plan_run: no stream table");
    const int prc = plan_patch(ops, n_ops, patches, n_patches, slots, n_slots);
    if (prc != ATMVFI_OK) return prc;
    for (int i = 0; i < n_ops; ++i) {
        const atmvfi_plan_op* op = ops + i;
        const int lane = lanes ? lanes[i] : 0;
        if (lane < 0 || lane >= n_streams) {
            if (failed_op) *failed_op = i;
            atmvfi::set_error("plan_run: op %d on lane %d of %d", i, lane, n_streams);
            return ATMVFI_EINVAL;
        }
        if (op->fn == ATMVFI_PLAN_RECORD || op->fn == ATMVFI_PLAN_WAIT) {
            const int64_t e = op->a[0].i;
            if (op->nargs != 1 || !events || e < 0 || e >= n_events || !events[e]) {
                if (failed_op) *failed_op = i;
                atmvfi::set_error("plan_run: op %d: event %lld of %d", i, (long long)e, n_events);
                return ATMVFI_EINVAL;
            }
            const hipError_t he = op->fn == ATMVFI_PLAN_RECORD ? hipEventRecord((hipEvent_t)events[e], (hipStream_t)streams[lane])
                                                               : hipStreamWaitEvent((hipStream_t)streams[lane], (hipEvent_t)events[e], 0);
            if (he != hipSuccess) {
                if (failed_op) *failed_op = i;
                atmvfi::set_error("plan_run: op %d: %s: %s", i, op->fn == ATMVFI_PLAN_RECORD ? "hipEventRecord" : "hipStreamWaitEvent", hipGetErrorString(he));
                return ATMVFI_ELAUNCH;
            }
            continue;
        }
        if (op->fn < 0 || op->fn >= ATMVFI_PLAN_NFN || op->nargs != kPlanFnArgs[op->fn]) {
            if (failed_op) *failed_op = i;
            atmvfi::set_error("plan_run: op %d has function id %d with %d arguments", i, op->fn, op->nargs);
            return ATMVFI_EINVAL;
        }
        const int rc = plan_call(op, streams[lane]);
        if (rc != ATMVFI_OK) {
            if (failed_op) *failed_op = i;
            return rc;
        }
    }
    return ATMVFI_OK;
}
